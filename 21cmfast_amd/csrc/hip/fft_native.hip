// fft_native.hip -- hand-written power-of-two 3-D real FFT for gfx950 with the k-space
// filter fused into its first pass.
//
// Replaces, for power-of-two boxes, the reference's
//   memcpy -> filter_box -> dft_c2r_cube        (src/py21cmfast/src/IonisationBox.c:577-663,
//                                                 filtering.c:308-394, dft.c:18-44)
// by three HBM sweeps per grid (rocFFT needs three as well, but unfused and at ~0.75 TB/s):
//
//   pass X  lines along x (stride ny*nz/2), W(kR) applied while loading, src -> work
//   pass Y  lines along y (stride nz/2), in place on work
//   pass Z  complex-to-real along z (contiguous lines) via a half-length complex FFT,
//           work -> real padded/dense grid
//
// Internal "split" k-space layout (all complex float2):
//   main[nx][ny][nz/2]   the k_z = 0 .. nz/2-1 columns  -> rows of nz/2*8 B, 128-B aligned
//   nyq [nx][ny]         the k_z = nz/2 (Nyquist) plane, stored after main
// FFTW's in-place padded layout has rows of (nz/2+1) complex = odd length, which would
// misalign every row and straddle cache lines; splitting the Nyquist plane off keeps all
// tile accesses aligned 128-B segments and wastes no bytes.
//
// Each workgroup (256 threads = 4 wavefronts) owns a tile of TZ = 16 adjacent columns x
// the full line length N in LDS (N*128 B: 64 KB at N = 512, two workgroups per CU), and
// runs a Stockham autosort FFT on it with radix-8/4/2 register butterflies.  Lanes span
// the 16 columns first, so every LDS access of a 16-lane group is one contiguous 128-B
// row: bank-conflict free without padding.  Between stages the data stay in LDS; each
// stage is read-all / barrier / write-all, in place.
//
// Window evaluation (double sincos per k-cell, as the reference does) is the only
// non-trivial ALU work.  |k| is even in k_x and k_y, so a thread evaluates W once for the
// rows (kx, nx-kx) and the workgroup reuses it for the tile pair (ky, ny-ky): 1/4 of the
// evaluations of a naive sweep, bit-identical values.
#include <hip/hip_runtime.h>

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "c21hip.h"
#include "fcoll_device.h"
#include "ms_window.h"
#include "c21cm_abi.h"
#include "c21cm_grid.h"

// experiment switches (tools/build_variant.sh; the defaults are the shipped configuration)
#ifndef C21X_XPAIR_THREADS
#define C21X_XPAIR_THREADS 512
#endif
#ifndef C21X_FUSE512  // 512-point line passes: first and last radix-8 stage in registers (one LDS stage)
#define C21X_FUSE512 1
#endif
#ifndef C21X_FUSE1024  // the same for 1024-point lines (radix-2 split + the first stage of both halves in registers)
#define C21X_FUSE1024 1
#endif
#ifndef C21X_F512_MERGE   // 1: second stage of both tiles of a two-radius sweep between one pair of barriers
#define C21X_F512_MERGE 1
#endif
#ifndef C21X_F512_HOIST2  // 1: second-stage twiddles kept in registers for the whole kernel (12-30 spilled
#define C21X_F512_HOIST2 0  // VGPRs in the two-radius kernels, no faster)
#endif
#ifndef C21X_XPAIR_SKIP_FFT  // diagnostic (wrong results): 1 = the two-radius pass X skips its transforms (512-point
#define C21X_XPAIR_SKIP_FFT 0  // lines: windows + one LDS round trip stay), 3 = registers straight back out (R + 2 W only)
#endif
#ifndef C21X_XPAIR_TWO_SETS
#define C21X_XPAIR_TWO_SETS 0
#endif
#ifndef C21X_XPAIR_NOWIN
#define C21X_XPAIR_NOWIN 0
#endif
#ifndef C21X_ZW_OCC       // min waves per SIMD requested for the fused pass Z (0: compiler's choice)
#define C21X_ZW_OCC 0
#endif
#ifndef C21X_ZW_TSRC_OCC  // waves per SIMD asked of the three-line barrier kernel of the recombination loop
#define C21X_ZW_TSRC_OCC 1
#endif
#ifndef C21X_EPI4_MASK16
#define C21X_EPI4_MASK16 1
#endif
#ifndef C21X_ZW_TSRC_LEAN  // three-line barrier kernel: mask rows as 16-byte pieces, N_rec rows parked in LDS
#define C21X_ZW_TSRC_LEAN 1
#endif
#ifndef C21X_ZW_TSRC_FENCE
#define C21X_ZW_TSRC_FENCE 1
#endif
#ifndef C21X_ZW_BLOCK     // threads per workgroup of the fused pass Z on 512-point lines (64, 128, 256)
#define C21X_ZW_BLOCK 256
#endif
#ifndef C21X_ZW_DPP       // 1: the mirror partner X[H-k] of the c2r pre-processing through DPP lane moves
#define C21X_ZW_DPP 0     //    (16 lanes per line) instead of a round trip through LDS: 230 instead of 184
#endif                    //    VGPRs, fused pass Z 0.304 ms in the loop either way (354 against 315 us alone)
#ifndef C21X_ZW_MASK16    // 1: mask rows of the fused pass Z as 16-byte loads / stores through the line's LDS region (measured: 312 vs 306 us, off)
#define C21X_ZW_MASK16 0
#endif
#ifndef C21X_ZW_LATE      // 1: second grid and mask rows requested after the first transform
#define C21X_ZW_LATE 0
#endif

namespace {
constexpr int kBlock = 256;
constexpr int TZ = 16;  // columns per tile (128 B of float2) for lines up to 512 points
// 1024-point lines: a 16-column tile is 128 KB of LDS, which fits, but a direct 1024-point
// Stockham plan holds twice the values per thread between its LDS stages and spills.  The line
// transform is split instead (line_fft): one in-place radix-2 decimation-in-frequency stage, two
// 512-point transforms of the halves, and the even/odd interleave folded into the row index of
// the store; one register set instead of two keeps the kernel inside 256 VGPRs.
// 1536-point lines (the reference's default DIM = 3 HII_DIM at HII_DIM = 512): a 16-column tile
// would be 196 KB; 8 columns (64-byte row segments) are 98 KB and take the split transform of the
// 1024-point lines -- one radix-2 stage, two 768-point (= 3 x 2^8) transforms.
constexpr int line_tile_cols(int n) { return n == 1536 ? TZ / 2 : TZ; }
// LDS row that holds output index g of a line after line_fft
template <int N>
__device__ __forceinline__ int fft_out_row(int g) {
    return (N >= 1024) ? ((g & 1) ? N / 2 + (g >> 1) : (g >> 1)) : g;
}

// x-blocked split layout.  The main block is stored as [x / XB][y][x % XB][k_z] with
// XB = 2^xb_log2(nx): 1 (the plain [x][y][k_z]) up to 512-point x-lines, 8 for 1024.  At 1024^3 the
// rows of a pass-X tile would otherwise lie ny*nz/2*8 B = 4 MB apart, one page each, and 42 % of
// the pass's L1-TLB requests missed; blocked, 16 consecutive x are 4 KB apart and a tile
// touches 64 pages instead of 1024.  Memory line m = (x / XB * ny + y) * XB + x % XB holds the
// k_z row of the logical line x * ny + y; the Nyquist plane stays [x][y].
#ifndef C21X_XB_MIN   // experiment switches: shortest x-line stored blocked, log2 of the block
#define C21X_XB_MIN 1024
#endif
#ifndef C21X_XB_LOG2  // 8 planes per block: pass Y's rows lie 32 KB apart instead of 64 (437 against 447 ms per
#define C21X_XB_LOG2 3  // 1024^3 call; blocks of 4: 448, of 32: 467)
#endif
__host__ __device__ constexpr int split_xb_log2(int nx) { return nx >= C21X_XB_MIN ? C21X_XB_LOG2 : 0; }
__host__ __device__ __forceinline__ long logical_line(long m, int ny, int lb) {
    if (lb == 0) return m;
    const long blk = (long)ny << lb;
    const long xbk = m / blk, rem = m - xbk * blk;
    const long y = rem >> lb, xi = rem & ((1 << lb) - 1);
    return ((xbk << lb) + xi) * ny + y;
}

#define LAUNCH_CHECK()                                                                  \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            c21hip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                       \
            return C21CM_IO_ERROR;                                                      \
        }                                                                               \
    } while (0)

#include "fft_device.h"  // complex helpers, Dft<R, SIGN>, wave_fence, wave_c2r

// The tile lives in LDS as tile[point * ROW + column], ROW >= COLS.  One stage of radix R
// with `s` = product of the radices already applied (log2s its log):
//   butterfly b in [0, N/R): inputs  tile[b + k*N/R],          k = 0..R-1
//                            outputs tile[q + s*R*p + s*j] *= tw[p*s*j],  p = b/s, q = b%s
// `tw` holds exp(-2 pi i t / N), t = 0..N-1 (conjugated for SIGN > 0).
template <int N, int COLS, int ROW, int R, int SIGN, bool LAST, int THREADS>
__device__ __forceinline__ void stockham_stage(float2 *tile, const float2 *tw, int log2s) {
    constexpr int NB = N / R;
    constexpr int ITEMS = NB * COLS;
    constexpr int kBlock = THREADS;
    constexpr int PER = (ITEMS + kBlock - 1) / kBlock;
    float2 v[PER][R];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = threadIdx.x + kBlock * u;
        if (ITEMS % kBlock == 0 || i < ITEMS) {
            const int col = i % COLS, b = i / COLS;
#pragma unroll
            for (int k = 0; k < R; k++) v[u][k] = tile[(b + k * NB) * ROW + col];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = threadIdx.x + kBlock * u;
        if (ITEMS % kBlock == 0 || i < ITEMS) {
            const int col = i % COLS, b = i / COLS;
            const int p = b >> log2s, q = b & ((1 << log2s) - 1);
            const int ps = p << log2s;
            Dft<R, SIGN>::run(v[u]);
            const int base = q + ((R * p) << log2s);
            // twiddles tw[ps*j], j = 1..R-1: three table reads (j = 1, 2, 4), the rest as
            // single products of exact table values
            float2 wj[R];
            wj[0] = make_float2(1.f, 0.f);
            if (!LAST && R >= 2) wj[1] = tw[ps];
            if (!LAST && R >= 4) {
                wj[2] = tw[ps * 2];
                wj[3] = cmul(wj[1], wj[2]);
            }
            if (!LAST && R >= 8) {
                wj[4] = tw[ps * 4];
                wj[5] = cmul(wj[1], wj[4]);
                wj[6] = cmul(wj[2], wj[4]);
                wj[7] = cmul(wj[3], wj[4]);
            }
#pragma unroll
            for (int j = 0; j < R; j++) {
                float2 o = v[u][j];
                if (!LAST && j > 0) {  // the final stage has p = 0: all twiddles are 1
                    float2 w = wj[j];
                    if (SIGN > 0) w.y = -w.y;
                    o = cmul(o, w);
                }
                tile[(base + (j << log2s)) * ROW + col] = o;
            }
        }
    }
    __syncthreads();
}

// Full length-N transform of every column of the tile, N = 2^L or 3 * 2^L: radix plan
// 8,8,..,{4,2} and, for the factor 3 of the reference's default DIM = 3 HII_DIM grids, one radix-3
// stage LAST -- every earlier stage then has a power-of-two stride s (shift / mask indexing), and
// the last stage of a Stockham plan has p = 0: no twiddles, outputs in place at b + j N/3.
template <int N, int COLS, int ROW, int SIGN, int THREADS>
__device__ __forceinline__ void fft_tile(float2 *tile, const float2 *tw) {
    constexpr int L = __builtin_ctz(N);
    constexpr int ODD = N >> L;
    static_assert((ODD == 1 || ODD == 3) && N >= 8, "line length 2^L or 3 * 2^L");
    int log2s = 0;
    constexpr int N8 = L / 3;   // radix-8 stages
    constexpr int REM = L % 3;  // 0, 1 (radix 2) or 2 (radix 4)
    constexpr bool P2LAST = (ODD == 1);  // the power-of-two stages end the plan
#pragma unroll
    for (int st = 0; st < N8; st++) {
        if (P2LAST && REM == 0 && st == N8 - 1)
            stockham_stage<N, COLS, ROW, 8, SIGN, true, THREADS>(tile, tw, log2s);
        else
            stockham_stage<N, COLS, ROW, 8, SIGN, false, THREADS>(tile, tw, log2s);
        log2s += 3;
    }
    if (REM == 1) stockham_stage<N, COLS, ROW, 2, SIGN, P2LAST, THREADS>(tile, tw, log2s);
    if (REM == 2) stockham_stage<N, COLS, ROW, 4, SIGN, P2LAST, THREADS>(tile, tw, log2s);
    if (ODD == 3) stockham_stage<N, COLS, ROW, 3, SIGN, true, THREADS>(tile, tw, L);
}

// Line transform of a whole tile.  N < 1024: the Stockham plan, natural order out.  N = 1024:
//   a[k] = x[k] + x[k+512],  b[k] = (x[k] - x[k+512]) w^k   (in place, rows k and k + 512)
//   X[2m] = FFT512(a)[m] -> row m,   X[2m+1] = FFT512(b)[m] -> row 512 + m
// (fft_out_row gives the row of an output index).  tw: exp(-2 pi i t / N); tw_half: the
// 512-point table exp(-2 pi i t / 512) = tw[2 t], only read when N = 1024.
template <int N, int COLS, int SIGN, int THREADS>
__device__ __forceinline__ void line_fft(float2 *tile, const float2 *tw, const float2 *tw_half) {
    if constexpr (N < 1024) {
        fft_tile<N, COLS, COLS, SIGN, THREADS>(tile, tw);
    } else {
        constexpr int HN = N / 2;
        for (int i = threadIdx.x; i < HN * COLS; i += THREADS) {
            const int col = i % COLS, k = i / COLS;
            const float2 x0 = tile[k * COLS + col], x1 = tile[(k + HN) * COLS + col];
            float2 w = tw[k];
            if (SIGN > 0) w.y = -w.y;
            tile[k * COLS + col] = cadd(x0, x1);
            tile[(k + HN) * COLS + col] = cmul(csub(x0, x1), w);
        }
        __syncthreads();
        fft_tile<HN, COLS, COLS, SIGN, THREADS>(tile, tw_half);
        fft_tile<HN, COLS, COLS, SIGN, THREADS>(tile + HN * COLS, tw_half);
    }
}

// ------------------------------------------------------------------ window functions
// reference: filtering.c:18-32, 80-117; identical to grid_kernels.hip (kept in this TU so the
// compiler can inline them into the fused pass).
struct ExpMfpConsts {
    double R, ratio, ratio2, ratio3, exp_term, ts_0, ts_2;
};
struct FilterParams {
    int type;  // -1: no filter
    int libm_trig;  // 1: general-range library sincos instead of fast_sincos (A/B switch)
    float R, R_param;
    double dkx, dky, dkz;
    ExpMfpConsts mfp;
};

// sin and cos of a non-negative argument below ~1e6 (kR never exceeds a few hundred here):
// two-constant Cody-Waite reduction with FMA (r = x - n*pi/2 to < 1 ulp) followed by the
// classic minimax kernels on [-pi/4, pi/4].  Results agree with a correctly rounded libm
// to ~1 ulp (double) at about a quarter of the instructions of the general-range routine,
// whose huge-argument path these kernels never need.
__device__ __forceinline__ void fast_sincos(double x, double *sn, double *cs) {
    const double n = rint(x * 6.36619772367581382433e-01);  // 2/pi
    double r = fma(-n, 1.57079632679489655800e+00, x);       // pi/2 high
    r = fma(-n, 6.12323399573676603587e-17, r);              // pi/2 low
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 +
                      z * (-1.98412698298579493134e-04 +
                           z * (2.75573137070700676789e-06 +
                                z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = z * (4.16666666666666019037e-02 +
                           z * (-1.38888888888741095749e-03 +
                                z * (2.48015872894767294178e-05 +
                                     z * (-2.75573143513906633035e-07 +
                                          z * (2.08757232129817482790e-09 +
                                               z * -1.13596475577881948265e-11)))));
    const double c = 1.0 - (0.5 * z - z * pc);
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c : s;
    const double cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ void sincos_sel(int libm, double x, double *s, double *c) {
    if (libm)
        sincos(x, s, c);
    else
        fast_sincos(x, s, c);
}

__device__ __forceinline__ double w_shell(double k, double R_inner, double R_outer, int libm) {
    const double kRi = k * R_inner, kRo = k * R_outer;
    if (kRo < 1e-4) {
        const double q = R_inner / R_outer;
        const double q3 = q * q * q;
        return 1. - kRo * kRo / 10 * (q3 * q * q - 1) / (q3 - 1);
    }
    double si, ci, so, co;
    sincos_sel(libm, kRi, &si, &ci);
    sincos_sel(libm, kRo, &so, &co);
    return 3.0 / (kRo * kRo * kRo - kRi * kRi * kRi) * (so - co * kRo - si + ci * kRi);
}
__device__ __forceinline__ float k_of(int n, int dim, double dk) {
    return (n > dim / 2) ? (float)((double)(n - dim) * dk) : (float)((double)n * dk);
}
// filtering.c:357-361 (real-space top-hat) and :80-104 (top-hat x exp(-r/mfp)) given sin/cos
__device__ __forceinline__ double w_tophat(double kR, double sn, double cs) {
    return (kR < 1e-4) ? 1 - kR * kR / 10 : 3.0 / (kR * kR * kR) * (sn - cs * kR);
}
__device__ __forceinline__ double w_expmfp(const ExpMfpConsts &c, double kR, double sn, double cs) {
    double f = (kR * kR * c.ratio2 + 2 * c.ratio + 1) * c.ratio * cs;
    f += (kR * kR * (c.ratio2 - c.ratio3) + c.ratio + 1) * sn / kR;
    f *= c.exp_term;
    f -= 2 * c.ratio2;
    const double d = kR * c.ratio * kR * c.ratio + 1;
    f *= -3 * c.ratio / (d * d);
    return (kR < 1e-4) ? c.ts_0 + c.ts_2 * kR * kR : f;
}
// Window values for NE modes at once.  The filter-type switch is hoisted out of the
// per-mode work and every step is written as a loop over the NE independent values, so
// the fp64 dependency chains (sqrt -> reduce -> polynomial -> divide) of different modes
// interleave instead of running back to back: with only two waves per SIMD in the line
// pass, instruction-level parallelism is what hides the fp64 latency.
template <int NE>
__device__ __forceinline__ void window_batch(const FilterParams &p, const float (&kx)[NE],
                                             const float (&ky)[NE], const float (&kz)[NE],
                                             double (&w)[NE]) {
    float ksq[NE];
#pragma unroll
    for (int i = 0; i < NE; i++)
        ksq[i] = __fadd_rn(__fadd_rn(__fmul_rn(kx[i], kx[i]), __fmul_rn(ky[i], ky[i])),
                           __fmul_rn(kz[i], kz[i]));
    if (p.type == 2) {  // Gaussian: kR^2 held in float (filtering.c:369)
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const float kR = __fmul_rn(__fmul_rn(ksq[i], p.R), p.R);
            w[i] = exp(-0.643 * 0.643 * (double)kR / 2.);
        }
        return;
    }
    double k[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) k[i] = sqrt((double)ksq[i]);
    if (p.type == 4) {
#pragma unroll
        for (int i = 0; i < NE; i++)
            w[i] = w_shell(k[i], (double)p.R, (double)p.R_param, p.libm_trig);
        return;
    }
    // types 0, 1 hold kR in float (filtering.c:331,357,364); type 3 keeps it in double (:83)
    double x[NE];
#pragma unroll
    for (int i = 0; i < NE; i++)
        x[i] = (p.type == 3) ? k[i] * p.mfp.R : (double)(float)(k[i] * (double)p.R);
    if (p.type == 1) {
#pragma unroll
        for (int i = 0; i < NE; i++) w[i] = (x[i] * 0.413566994 > 1) ? 0. : 1.;
        return;
    }
    double sn[NE], cs[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) sincos_sel(p.libm_trig, x[i], &sn[i], &cs[i]);
    if (p.type == 0) {
#pragma unroll
        for (int i = 0; i < NE; i++) w[i] = w_tophat(x[i], sn[i], cs[i]);
    } else {
#pragma unroll
        for (int i = 0; i < NE; i++) w[i] = w_expmfp(p.mfp, x[i], sn[i], cs[i]);
    }
}

// Two windows of the same modes (the density grid's HII_FILTER and the emissivity grid's
// exp-MFP filter of one radius).  Top-hat + exp-MFP share |k| and one sincos: the top-hat
// argument is kR rounded to float (filtering.c:357), the exp-MFP one is kR in double (:83),
// so sin/cos of the second follow from the first by a third-order rotation through their
// difference d <= 2^-24 kR (error d^4/24 < 1e-20 for kR of a few hundred).
template <int NE>
__device__ __forceinline__ void window_batch_dual(const FilterParams &pa, const FilterParams &pb,
                                                  const float (&kx)[NE], const float (&ky)[NE],
                                                  const float (&kz)[NE], double (&wa)[NE],
                                                  double (&wb)[NE]) {
    if (!(pa.type == 0 && pb.type == 3 && !pa.libm_trig)) {
        window_batch<NE>(pa, kx, ky, kz, wa);
        window_batch<NE>(pb, kx, ky, kz, wb);
        return;
    }
    double x0[NE], x3[NE], sn[NE], cs[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) {
        const float ksq = __fadd_rn(__fadd_rn(__fmul_rn(kx[i], kx[i]), __fmul_rn(ky[i], ky[i])),
                                    __fmul_rn(kz[i], kz[i]));
        const double k = sqrt((double)ksq);
        x0[i] = (double)(float)(k * (double)pa.R);
        x3[i] = k * pb.mfp.R;
    }
#pragma unroll
    for (int i = 0; i < NE; i++) fast_sincos(x0[i], &sn[i], &cs[i]);
#pragma unroll
    for (int i = 0; i < NE; i++) {
        wa[i] = w_tophat(x0[i], sn[i], cs[i]);
        const double d = x3[i] - x0[i], d3 = d * (1.0 / 3.0);
        const double s3 = sn[i] + d * (cs[i] - 0.5 * d * (sn[i] + d3 * cs[i]));
        const double c3 = cs[i] - d * (sn[i] + 0.5 * d * (cs[i] - d3 * sn[i]));
        wb[i] = w_expmfp(pb.mfp, x3[i], s3, c3);
    }
}

// ------------------------------------------------------------------ window table of one radius
// W(kR) depends on (|k_x|, |k_y|, k_z) only, so one radius needs (nx/2+1)(ny/2+1)(nz/2+1)
// values -- a quarter of the modes pass X multiplies.  They are evaluated here, at full
// occupancy, into
//     main [|k_y|][|k_x|][k_z < nz/2]   and   nyq [|k_x|][|k_y|]   (k_z = nz/2),
// laid out so that pass X streams them exactly like its data tiles (rows of 16 k_z values =
// 128 contiguous bytes).  An earlier version evaluated W inside pass X (once per mirror pair,
// kept in LDS): the same total time at 512^3, but 28 KB more LDS and 1024-thread workgroups,
// and no room for a 1024-point tile.
// Table entries are stored as float: W is evaluated in double and rounded once.  The reference
// multiplies the float spectrum by the double window ((float)(v * W), 0.5 ulp); with the rounded
// window the product carries at most one more half-ulp (6e-8 relative, well below the rounding
// of the transforms themselves), the table traffic -- 0.27 of the 0.54 GB a radius wrote and
// read back for its two windows at 512^3 -- halves, and pass X multiplies in fp32.
using wtab_t = float;
struct WTableArgs {
    FilterParams pa, pb;
    int dual;  // 1: also fill the b tables with window pb
    int nx, ny, nz;
    wtab_t *main_a, *nyq_a, *main_b, *nyq_b;
    MsConsts ms_a, ms_b;  // windows of type 5 (multiple scattering), MS kernel variant only
};

// type 5 keeps |k|^2 in float like every other window (filtering.c:347-381)
template <int NE>
__device__ __forceinline__ void window_batch_ms(const FilterParams &p, const MsConsts &ms,
                                                const float (&kx)[NE], const float (&ky)[NE],
                                                const float (&kz)[NE], double (&w)[NE]) {
    if (p.type != 5) {
        window_batch<NE>(p, kx, ky, kz, w);
        return;
    }
#pragma unroll
    for (int i = 0; i < NE; i++) {
        const float ksq = __fadd_rn(__fadd_rn(__fmul_rn(kx[i], kx[i]), __fmul_rn(ky[i], ky[i])),
                                    __fmul_rn(kz[i], kz[i]));
        w[i] = ms_window(sqrt((double)ksq), ms);
    }
}

// MS = true: the variant that knows the multiple-scattering window; kept apart so that its
// series loop does not weigh on the registers of the excursion-set loop's table kernel.
template <bool MS>
__global__ void __launch_bounds__(kBlock)
window_table_kernel(WTableArgs t) {
    const int nxh = t.nx / 2 + 1, nyh = t.ny / 2 + 1, H = t.nz / 2, H4 = H / 4;
    const long n4 = (long)nyh * nxh * H4;
    const long id = (long)blockIdx.x * kBlock + threadIdx.x;
    // nx == ny (x and y share the box length): W is symmetric in (|k_x|, |k_y|), so only
    // i <= j is evaluated and written to both places (wave-uniform: a wave shares (i, j))
    const bool sym = (t.nx == t.ny);
    if (id < n4) {
        const int l4 = (int)(id % H4);
        const long r = id / H4;
        const int i = (int)(r % nxh), j = (int)(r / nxh);
        if (sym && i > j) return;
        const long rT = (long)i * nxh + j;  // the mirrored row [i][j]
        float kx[4], ky[4], kz[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            kx[e] = k_of(i, t.nx, t.pa.dkx);
            ky[e] = k_of(j, t.ny, t.pa.dky);
            kz[e] = (float)((double)(4 * l4 + e) * t.pa.dkz);
        }
        double wa[4], wb[4];
        if (MS) {
            window_batch_ms<4>(t.pa, t.ms_a, kx, ky, kz, wa);
            if (t.dual) window_batch_ms<4>(t.pb, t.ms_b, kx, ky, kz, wb);
        } else if (t.dual)
            window_batch_dual<4>(t.pa, t.pb, kx, ky, kz, wa, wb);
        else
            window_batch<4>(t.pa, kx, ky, kz, wa);
        for (int rep = 0; rep < ((sym && i != j) ? 2 : 1); rep++) {
            const long rr = rep ? rT : r;
            *reinterpret_cast<float4 *>(t.main_a + rr * H + 4 * l4) =
                make_float4((float)wa[0], (float)wa[1], (float)wa[2], (float)wa[3]);
            if (t.dual)
                *reinterpret_cast<float4 *>(t.main_b + rr * H + 4 * l4) =
                    make_float4((float)wb[0], (float)wb[1], (float)wb[2], (float)wb[3]);
        }
    } else if (id - n4 < (long)nxh * nyh) {
        const long q = id - n4;
        const int i = (int)(q / nyh), j = (int)(q % nyh);
        if (sym && i > j) return;
        float kx[1] = {k_of(i, t.nx, t.pa.dkx)}, ky[1] = {k_of(j, t.ny, t.pa.dky)};
        float kz[1] = {(float)((double)(t.nz / 2) * t.pa.dkz)};
        double wa[1], wb[1];
        if (MS) {
            window_batch_ms<1>(t.pa, t.ms_a, kx, ky, kz, wa);
            if (t.dual) window_batch_ms<1>(t.pb, t.ms_b, kx, ky, kz, wb);
        } else if (t.dual)
            window_batch_dual<1>(t.pa, t.pb, kx, ky, kz, wa, wb);
        else
            window_batch<1>(t.pa, kx, ky, kz, wa);
        t.nyq_a[q] = (wtab_t)wa[0];
        if (t.dual) t.nyq_b[q] = (wtab_t)wb[0];
        if (sym && i != j) {
            t.nyq_a[(long)j * nyh + i] = (wtab_t)wa[0];
            if (t.dual) t.nyq_b[(long)j * nyh + i] = (wtab_t)wb[0];
        }
    }
}

// ------------------------------------------------------------------ node tables of W(x)
// The windows of the excursion-set loop are functions of x = kR alone: the real-space top-hat
// (filtering.c:357-361; the SAME function for every radius) and the top-hat x exp(-r/mfp) window
// (:80-104; one function per radius, through R/mfp).  Instead of a 3-D table of W per radius
// (0.14 GB per window, written by one kernel and streamed by pass X: 10-14 % of the R loop's HBM
// traffic at 512^3) pass X can interpolate W from nodes 1/4 apart held in LDS: per node
// (W, W' h, W'' h^2/2), quintic Hermite between neighbours.  This kernel fills the node tables of
// one call: table 0 = top-hat, tables 1.. = the exp-MFP window of each radius.  Values in double
// exactly as filter_box evaluates them; the derivatives by 8th-order central differences with step
// h/16 (both windows are even in x; truncation ~1e-17, rounding ~1e-13 of the envelope).
struct WNodeArgs {
    float *out;        // [n_tabs][n_nodes][3]
    int n_nodes, n_tabs;
    int first_type;    // type of table 0 (0 or 3); tables 1.. are type 3
    const ExpMfpConsts *mfp;  // device array, one per type-3 table in table order
};
__device__ __forceinline__ double wnode_value(int type, const ExpMfpConsts &c, double x) {
    x = fabs(x);
    double sn, cs;
    sincos(x, &sn, &cs);
    return type == 0 ? w_tophat(x, sn, cs) : w_expmfp(c, x, sn, cs);
}
__global__ void __launch_bounds__(kBlock)
window_nodes_kernel(WNodeArgs a) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= a.n_nodes * a.n_tabs) return;
    const int tab = id / a.n_nodes, n = id - tab * a.n_nodes;
    const int type = (tab == 0) ? a.first_type : 3;
    ExpMfpConsts c{};
    if (type == 3) c = a.mfp[a.first_type == 3 ? tab : tab - 1];
    const double h = 0.25, s = h / 16., x = h * (double)n;
    const double c1[4] = {4. / 5., -1. / 5., 4. / 105., -1. / 280.};
    const double c2[4] = {8. / 5., -1. / 5., 8. / 315., -1. / 560.};
    const double f0 = wnode_value(type, c, x);
    double d1 = 0., d2 = -205. / 72. * f0;
#pragma unroll
    for (int k = 1; k <= 4; k++) {
        const double fp = wnode_value(type, c, x + k * s), fm = wnode_value(type, c, x - k * s);
        d1 += c1[k - 1] * (fp - fm);
        d2 += c2[k - 1] * (fp + fm);
    }
    float *o = a.out + 3 * (size_t)id;
    o[0] = (float)f0;
    o[1] = (float)(d1 / s * h);
    o[2] = (float)(d2 / (s * s) * h * h * 0.5);
}

// ------------------------------------------------------------------ pass X / pass Y
// One launch covers up to two GEOMETRIES (the main block [nx][ny][nz/2] and the k_z = nz/2
// Nyquist plane [nx][ny]) and up to two GRIDS of identical shape (density and emissivity
// spectra), so a filter radius costs one pass-X and one pass-Y launch, and the window
// W(kR) -- the same for both grids -- is evaluated once per mode.
struct LineGeo {
    const float2 *src[2];  // per grid
    float2 *dst[2];
    float2 *dst2[2];       // FMODE 5: the second radius' outputs
    // offset of index i along the line / the outer axis: (i >> lb) * bstride + (i & (2^lb - 1)) *
    // stride; lb = 0 (unblocked): i * bstride
    long line_stride, line_bstride;
    long outer_stride, outer_bstride;
    int line_lb, outer_lb;
    long col_stride;    // elements between adjacent columns of a tile (1 = vector loads)
    int n_outer;        // outer index count (tiles along the non-transformed, non-column axis)
    int n_ctiles;       // column tiles (columns / TZ)
    int pair_outer;     // 1: a workgroup handles the mirror pair (o, n_outer - o)
    int filter_axis;    // 0: columns are k_z, outer is k_y (main block)
                        // 1: columns are k_y, k_z fixed at nz/2 (Nyquist plane)
};

struct LinePassArgs {
    LineGeo g0, g1;
    int n_geo;    // 1: g0 only, 2: g0 then g1
    int g1_strided;  // g1 has col_stride != 1 (y-lines of the Nyquist plane): epilogue tiles
    int n_grids;  // 1 or 2
    int n_y, n_z;  // grid dims for the wavenumbers
    float out_scale;  // applied at store (1 = none)
    // FMODE 3: window tables of this radius (window_table_kernel), per window
    const wtab_t *wt_main[2];  // [ny/2+1][nx/2+1][nz/2]
    const wtab_t *wt_nyq[2];   // [nx/2+1][ny/2+1]
    int dual;                  // FMODE 3: grid 1 uses window 1 (else both grids use window 0)
    // FMODE 5 (pass X only): two radii per sweep -- each source tile is read once, multiplied
    // by the windows of radius A and transformed into dst, then by those of radius B into dst2
    const wtab_t *wt2_main[2];
    const wtab_t *wt2_nyq[2];
    FilterParams fp;           // the window of grid 0 (host side: table construction)
    // FMODE 4 (pass X only): separable k-space operator applied on load,
    //   v *= sign * k_x^ex * k_y^ey * k_z^ez  (times i when imag), k_a = index_to_k(i_a) in double:
    // the gradient / second-derivative operators of the ICs on a spectrum pre-divided by k^2
    // (InitialConditions.c:240-297).  Every factor is one-dimensional: a row scalar (k_x), a
    // tile scalar (k_y) and a column scalar (k_z), so nothing is looked up.
    int op_ex, op_ey, op_ez, op_imag;
    double op_sign, op_dkx, op_dky, op_dkz;
    // FMODE 6 / 7 (pass X): windows evaluated IN the kernel from node tables of W(x), x = kR, held
    // in LDS (window_nodes_kernel): no 3-D table is built, written or streamed.  wev_src[t]: the
    // node tables staged into LDS (wev_n_nodes nodes of 3 floats each); wev_tab[radius][window]:
    // which staged table serves (sweep member, window), -1 = sharp-k (evaluated directly).
    const float *wev_src[3];
    int wev_n_tabs, wev_n_nodes;
    int wev_tab[2][2];
    int wev_type[2];   // window type of grid 0 / grid 1 (0 top-hat, 1 sharp-k, 3 exp-MFP)
    float wev_R[2];    // filter radius of sweep member 0 / 1
    // exp-MFP constants (ratio, ratio^2, ratio^3, exp(-1/ratio)) of sweep member 0 / 1: used beyond
    // the node tables' range (x >= (wev_n_nodes - 2) / 4), where the window is evaluated directly
    float wev_mfp[2][4];
    double wev_dkx, wev_dky, wev_dkz;
    // Item order of the main geometry (round 5 experiment): 0 = outer-major (k_y, then
    // the k_z column tiles).  Pass X, 1: column-tile-major -- the LAST items written are then whole
    // (x, column tile) tiles of pass Y.  Pass Y, k > 0: the last k column tiles first (all x, the tile
    // written last first), the rest outer-major as before (so that the fused pass Z still finds the
    // planes written last).
    int item_order;
};

static inline int geo_items(const LineGeo &g) {
    return (g.pair_outer ? (g.n_outer / 2 + 1) : g.n_outer) * g.n_ctiles;
}

// Tile loader geometry: thread t owns the float4 column pair c4 = t % 8 of the row pairs
//   row_a = r0 + 32u  in [0, N/2)   and its mirror   row_b = N - row_a  (N/2 when row_a = 0),
// r0 = t / 8, u = 0 .. N/64-1.  Mirror rows share |k_x|, hence the window value.
template <int N>
__device__ __forceinline__ int mirror_row(int row_a) {
    return row_a == 0 ? N / 2 : N - row_a;
}

// Persistent workgroups: each loops over (outer group, column tile) work items with a
// grid stride; a work item is the group's tiles (mirror pair x grids).  Within the loop the
// global loads of the NEXT tile are issued before the LDS transform of the current one, and
// the window values are computed while the first tile's loads are in flight, so HBM latency
// hides behind the LDS/ALU phase.
// THREADS = 512 at N >= 128 (one workgroup per CU at N >= 256, 2 waves per SIMD, <= 256
// VGPRs: room for the two register sets), else 256.
template <int N, int FMODE = 0>
struct LineThreads {
    // (the loader needs N/2 = rows per sweep x row pairs per thread: 192-point lines take 256;
    //  1024 threads for 1024-point lines spill 13-38 VGPRs at the 128-register budget: DESIGN 8.1)
    static constexpr int value =
        ((FMODE == 5 || FMODE == 7 || FMODE == 9) && N == 512) ? C21X_XPAIR_THREADS
                                                 : ((N >= 128 && N != 192) ? 512 : 256);
};

// the geometry of one work item, in wave-uniform registers
struct LineItemSecond {  // FMODE 5: the second radius' outputs and window tables
    float2 *dst0b, *dst1b;
    const wtab_t *wt0b, *wt1b;
};
struct LineItemNone {};
template <bool PAIR>
struct LineItemT : std::conditional_t<PAIR, LineItemSecond, LineItemNone> {
    const float2 *src0, *src1;
    float2 *dst0, *dst1;
    long line_stride, line_bstride, outer_stride, outer_bstride, col_stride;
    int line_lb, outer_lb;
    int n_outer, filter_axis;
    int og, ct, npair;
    const wtab_t *wt0, *wt1;  // FMODE 3: window tables of this geometry
};

// FMODE: 0 no window, 3 window streamed from the per-radius tables, 4 separable k-space
// operator (pass X of the IC transforms), 5 = 3 for two radii at once (one read, two outputs)
template <int N, int SIGN, int FMODE>
__global__ void __launch_bounds__((LineThreads<N, FMODE>::value), (LineThreads<N, FMODE>::value / 256))
line_pass_kernel(LinePassArgs a, const float2 *__restrict__ tw_global) {
    static_assert(N >= 64, "tile loader needs N >= 64");
    constexpr int kBlock = LineThreads<N, FMODE>::value;
    constexpr int TZ = line_tile_cols(N);  // shadows the namespace constant: this kernel's tile
    constexpr int CPAIR = TZ / 2;          // float4 (column pairs) per row
    constexpr int RSTEP = kBlock / CPAIR;  // rows covered by one sweep of the workgroup
    constexpr bool WEVAL = (FMODE >= 6 && FMODE <= 9);  // windows from node tables in LDS
    constexpr bool WDIRECT = (FMODE == 8 || FMODE == 9);  // ... and directly beyond the tables' range
    constexpr bool WIN = (FMODE == 3 || (FMODE == 5 && !C21X_XPAIR_NOWIN) || WEVAL);
    constexpr bool PAIR = (FMODE == 5 || FMODE == 7 || FMODE == 9);
    constexpr int NR = PAIR ? 2 : 1;     // radii per sweep
    using LineItem = LineItemT<PAIR>;
    extern __shared__ float4 lds_raw[];
    float2 *tile = reinterpret_cast<float2 *>(lds_raw);  // [N][TZ] (x 2 radii with FMODE 5)
    float2 *tw = tile + NR * N * TZ;                     // [N]
    float2 *tw_half = tw + N;                            // [N/2], N = 1024 only
    float *wnodes = reinterpret_cast<float *>(tw_half + (N >= 1024 ? N / 2 : 0));  // WEVAL
    for (int t = threadIdx.x; t < N; t += kBlock) tw[t] = tw_global[t];
    if (N >= 1024)
        for (int t = threadIdx.x; t < N / 2; t += kBlock) tw_half[t] = tw_global[2 * t];
    if constexpr (WEVAL) {
        const int per = 3 * a.wev_n_nodes;
        for (int t = threadIdx.x; t < per; t += kBlock) {
            wnodes[t] = a.wev_src[0][t];
            if (a.wev_n_tabs > 1) wnodes[per + t] = a.wev_src[1][t];
            if (a.wev_n_tabs > 2) wnodes[2 * per + t] = a.wev_src[2][t];
        }
    }
    // The twiddles (and node tables) are in LDS before anybody reads them.  Round 5: this barrier was
    // missing where neither the evaluated windows nor the 512-point path brought their own -- pass Y and the
    // forward passes of 1024-point lines read tw / tw_half for the register stage of a workgroup's FIRST
    // tile, which skips the loop's barrier: a wave that ran ahead of the wave that loads an entry read
    // stale LDS.  Invisible in isolation (the waves of a workgroup start together), it showed as one wrong
    // x-plane in about one 1024^3 call in thirty once four processes shared the GPU.
    // (C21X_NO_TW_BARRIER=1: the bug back in, to validate tests/test_gpu_contention.py against it)
#ifndef C21X_NO_TW_BARRIER
#define C21X_NO_TW_BARRIER 0
#endif
    if (!C21X_NO_TW_BARRIER || WEVAL) __syncthreads();

    constexpr int NP = (N / 2) / RSTEP;  // row pairs per thread
    const int r0 = threadIdx.x / CPAIR, c4 = threadIdx.x % CPAIR;
    // F512: the butterfly index of this thread is r0 in all three stages, so the twiddles of the
    // first (tw[r0 j]) and second stage (tw[(r0 & ~7) j]) are constants of the kernel
    constexpr bool F512_ = C21X_FUSE512 && N == 512 && kBlock == 512 && (FMODE == 0 || WEVAL);
    float2 twd1[F512_ ? 8 : 1], twd2[F512_ ? 8 : 1];
    if constexpr (F512_) {
        __syncthreads();  // tw is in LDS
        auto fill = [&](float2(&t)[8], int ps) {
            t[0] = make_float2(1.f, 0.f);
            t[1] = tw[ps];
            t[2] = tw[2 * ps];
            t[4] = tw[4 * ps];
            t[3] = cmul(t[1], t[2]);
            t[5] = cmul(t[1], t[4]);
            t[6] = cmul(t[2], t[4]);
            t[7] = cmul(t[3], t[4]);
            if (SIGN > 0)
#pragma unroll
                for (int j = 1; j < 8; j++) t[j].y = -t[j].y;
        };
        fill(twd1, r0);
        if (C21X_F512_HOIST2) fill(twd2, r0 & ~7);
    }
    const int n_work0 = (a.g0.pair_outer ? (a.g0.n_outer / 2 + 1) : a.g0.n_outer) * a.g0.n_ctiles;
    const int n_work1 =
        (a.n_geo > 1) ? (a.g1.pair_outer ? (a.g1.n_outer / 2 + 1) : a.g1.n_outer) * a.g1.n_ctiles : 0;
    // A strided g1 is handled after the pipelined loop, whose global loads and stores are all
    // 16-byte vectors in fixed numbers: mixing the two forms inside the loop made the
    // compiler wait with vmcnt(0) (i.e. also for the stores just issued) before every tile.
    const int n_work = n_work0 + (a.g1_strided ? 0 : n_work1);

    // explicit selects (not an indexed kernel-argument array, which would be copied to scratch)
    auto decode = [&](int w) {
        LineItem it;
        const bool s = w >= n_work0;
        int ww = s ? w - n_work0 : w;
        if (WIN && !s && ww < (n_work0 & ~15) && a.item_order == 0) {
            // XCD-aware order.  Workgroups are dealt to the 8 XCDs round robin, and two column
            // tiles of one k_y that are neighbours share every 128-byte line of the window table
            // (a tile reads 64 bytes per row).  Within each run of 16 items, workgroups b and
            // b + 8 -- same XCD, same L2 -- get such a pair, so the second request for a line
            // hits in that L2 instead of going to HBM a second time.
            const int r = ww & 15;
            ww = (ww & ~15) + 2 * (r & 7) + (r >> 3);
        }
        it.src0 = s ? a.g1.src[0] : a.g0.src[0];
        it.src1 = s ? a.g1.src[1] : a.g0.src[1];
        it.dst0 = s ? a.g1.dst[0] : a.g0.dst[0];
        it.dst1 = s ? a.g1.dst[1] : a.g0.dst[1];
        it.line_stride = s ? a.g1.line_stride : a.g0.line_stride;
        it.line_bstride = s ? a.g1.line_bstride : a.g0.line_bstride;
        it.line_lb = s ? a.g1.line_lb : a.g0.line_lb;
        it.outer_stride = s ? a.g1.outer_stride : a.g0.outer_stride;
        it.outer_bstride = s ? a.g1.outer_bstride : a.g0.outer_bstride;
        it.outer_lb = s ? a.g1.outer_lb : a.g0.outer_lb;
        it.col_stride = s ? a.g1.col_stride : a.g0.col_stride;
        it.n_outer = s ? a.g1.n_outer : a.g0.n_outer;
        it.filter_axis = s ? a.g1.filter_axis : a.g0.filter_axis;
        it.wt0 = (it.filter_axis == 1) ? a.wt_nyq[0] : a.wt_main[0];
        it.wt1 = (it.filter_axis == 1) ? a.wt_nyq[1] : a.wt_main[1];
        if constexpr (PAIR) {
            it.dst0b = s ? a.g1.dst2[0] : a.g0.dst2[0];
            it.dst1b = s ? a.g1.dst2[1] : a.g0.dst2[1];
            it.wt0b = (it.filter_axis == 1) ? a.wt2_nyq[0] : a.wt2_main[0];
            it.wt1b = (it.filter_axis == 1) ? a.wt2_nyq[1] : a.wt2_main[1];
        }
        const int nct = s ? a.g1.n_ctiles : a.g0.n_ctiles;
        const int po = s ? a.g1.pair_outer : a.g0.pair_outer;
        it.og = ww / nct;
        it.ct = ww - it.og * nct;
        if (a.item_order == 3 || a.item_order == -1) {
            // (only where a trip covers whole rows of column tiles, so that the rotation permutes the items
            //  of a trip: boxes with 6, 24 or 48 column tiles -- 192-, 768-, 1536-point z-lines -- keep the
            //  plain order)
            if (!s && (int)gridDim.x % nct == 0)
            // rotate: with 256 workgroups and 16 (32) column tiles a workgroup would meet the SAME column
            // tile on every trip (256 = 0 mod 16), and the tiles whose lines sit in the slow phase of the
            // HBM channel map always stall the same workgroups; trip j takes column tile (ct + j) instead
                it.ct = (it.ct + w / (int)gridDim.x) % nct;
        } else if (a.item_order != 0 && !s) {
            const int nog = n_work0 / nct;  // outer groups of the main geometry
            if (FMODE != 0) {               // pass X: column-tile-major
                it.ct = ww / nog;
                it.og = ww - it.ct * nog;
            } else {                        // pass Y: the last k column tiles first
                const int k = a.item_order < nct ? a.item_order : nct;
                const int head = k * nog;
                if (ww < head) {
                    const int c = ww / nog;
                    it.ct = nct - 1 - c;
                    it.og = ww - c * nog;
                } else {
                    const int v = ww - head, m = nct - k;
                    it.og = v / m;
                    it.ct = v - it.og * m;
                }
            }
        }
        it.npair = (po && it.og != 0 && 2 * it.og != it.n_outer) ? 2 : 1;
        return it;
    };
    // member m of an item: grid = m / npair, mirror index = m % npair
    auto member_grid = [](const LineItem &it, int m) { return it.npair == 2 ? (m >> 1) : m; };
    auto member_base = [](const LineItem &it, int m) {
        const int mi = it.npair == 2 ? (m & 1) : 0;
        const int outer = mi == 0 ? it.og : it.n_outer - it.og;
        return (long)(outer >> it.outer_lb) * it.outer_bstride +
               (long)(outer & ((1 << it.outer_lb) - 1)) * it.outer_stride +
               (long)it.ct * TZ * it.col_stride;
    };
    // element offset of row r (< N/2 + 1) of a line from the line's first point
    auto row_off = [](const LineItem &it, int r) {
        return (unsigned)(r >> it.line_lb) * (unsigned)it.line_bstride +
               (unsigned)(r & ((1 << it.line_lb) - 1)) * (unsigned)it.line_stride;
    };

    // Addressing: a wave-uniform 64-bit tile base (SGPRs) plus 32-bit per-thread element
    // offsets.  Rows < N/2 are addressed from the tile base, mirror rows from the row-N/2
    // base, so offsets stay below 2^31 elements even at 1024^3.
    // Two register sets: while tile t is transformed in LDS, the loads of tiles t+1 AND t+2
    // are in flight (a single set left HBM idle between a tile's arrival and the issue of the
    // next loads: 4.5 TB/s against the 6.3 TB/s a plain copy reaches).
    // Two radii per sweep: ONE set.  The next tile's loads go out before the two transforms of
    // this one, which is the look-ahead two sets buy the single-radius pass; the second set only
    // cost registers there (256 VGPRs + 84 bytes of scratch; 0.95 against 0.89 ms at 512^3).
    constexpr bool TWO_SETS = (N < 1024) && (!PAIR || C21X_XPAIR_TWO_SETS);
    float4 reg_a[2 * NP], reg_b[2 * NP];
    // FMODE 3: window values of this thread's row pairs x 2 columns; `pre` is in flight ahead
    // of the member that starts a new window, `cur` serves the members after it
    // Lines of 256 points and fewer load the next window straight into `cur` (it is dead between
    // a tile's LDS write and the next tile's): 10 % faster there (two workgroups per CU share the
    // register file), 3-6 % slower at 512 and 1024 points, which keep the separate `pre` set.
    // F512 (round 3): 512-point lines, no window or evaluated windows.  A thread owns the eight rows
    // b + 64 k of its two columns -- one complete input set of the first radix-8 Stockham stage and
    // one complete output set of the last -- so the first stage runs on the registers before the
    // tile goes to LDS and the last on the values read back for the store: ONE stage in LDS instead
    // of three.  Half the LDS traffic (256 instead of 512 KB per tile) and three barriers instead
    // of seven; LDS and global memory share a CU's data path to the registers, so the LDS bytes
    // were on the critical path (the two-radius pass X without its transforms ran at 5.0 TB/s,
    // with them at 3.6).  Mirror rows are no longer in one thread: eight window values per column
    // instead of four.
    constexpr bool F512 = C21X_FUSE512 && N == 512 && kBlock == 512 && (FMODE == 0 || WEVAL);
    // F1024: the same for 1024-point lines (16 rows r0 + 64 u per thread): the radix-2 split
    // a = x[k] + x[k + 512], b = (x[k] - x[k + 512]) w^k and the first stage of both 512-point halves
    // on the registers, the second stage of both halves in LDS, the third on the way out, where
    // A[m] = X[2 m] and B[m] = X[2 m + 1] go to adjacent rows.  Streamed window tables (FMODE 3) are
    // read per row (row and mirror row name the same table row; the partner thread's read hits L1).
    constexpr bool F1024 = C21X_FUSE1024 && N == 1024 && kBlock == 512 && (FMODE == 0 || FMODE == 3 || WEVAL);
    constexpr bool FUSED = F512 || F1024;
    constexpr int NW = FUSED ? 2 * NP : NP;  // window values (rows) per thread and column
    constexpr bool WPRE = (N >= 512) && !FUSED;
    float2 wpre[NR][WPRE ? NP : 1], wcur[NR][NW], wpre_half[NR], wcur_half[NR];
    auto w_reload = [&](const LineItem &it, int m) {
        const int mi = it.npair == 2 ? (m & 1) : 0;
        return m == 0 || (a.dual && mi == 0);
    };
    // WEVAL: window values from node tables, all in fp32 with two-float (hi + lo) steps where the
    // reference works in double.  |k| as filter_box forms it (filtering.c:347-352: float squares
    // summed in float, root in double): kh = sqrt_f32(ksq), kl = (ksq - kh^2) / (2 kh) with the
    // residual from one FMA -- kh + kl is the double root to ~1e-14.  x = k R: ph = fl(kh R),
    // e = (kh R - ph) + kl R (FMA residual), so x0 = fl(ph + e) is the reference's kR rounded to
    // float (top-hat and sharp-k, :331,357,364) and d = (ph - x0) + e its rounding residual: the
    // exp-MFP window takes kR in double (:83), i.e. the interpolant at x0 plus d times its slope.
    // Quintic Hermite on nodes 1/4 apart: within 1.2e-7 of the window's envelope of the double
    // evaluation (tools/window_interp_check.py).
    struct KAbs {
        float kh, kl;
    };
    auto k_abs = [](float kx, float ky, float kz) -> KAbs {
        const float ksq = __fadd_rn(__fadd_rn(__fmul_rn(kx, kx), __fmul_rn(ky, ky)), __fmul_rn(kz, kz));
        KAbs k;
        k.kh = __builtin_amdgcn_sqrtf(ksq);
        const float r = __fmaf_rn(-k.kh, k.kh, ksq);
        k.kl = (k.kh > 0.f) ? __fmul_rn(r, __fmul_rn(0.5f, __builtin_amdgcn_rcpf(k.kh))) : 0.f;
        return k;
    };
    // Beyond the node tables (large radii of the spin-temperature shells; the tables are sized by
    // the LDS left beside the tiles) the windows are evaluated directly in fp32: Cody-Waite
    // reduction of the float kR with two FMAs, minimax sin / cos, W = 3 (sin - x cos) / x^3 (and
    // the exp-MFP form with sin / cos rotated through the residual d).  There |W| < 3 / x^2 and
    // the result is within 5e-7 of that envelope (tools/window_interp_check.py).
    auto weval_direct = [&](float x0, float d, int type, int rr) -> float {
        const float n = rintf(x0 * 0.6366197723675814f);
        float r = __fmaf_rn(-n, 1.5707963705062866f, x0);
        r = __fmaf_rn(-n, -4.371138828673793e-08f, r);
        const float z = r * r;
        const float s = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
        const float c = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                                  4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
        const int q = (int)n & 3;
        const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
        float sn = (q & 2) ? -ss : ss, cs = ((q + 1) & 2) ? -cc : cc;
        const float inv = __builtin_amdgcn_rcpf(x0);
        if (type == 0) return 3.f * (sn - x0 * cs) * (inv * inv) * inv;
        const float s0 = sn, c0 = cs;
        sn = fmaf(d, c0, s0);
        cs = fmaf(-d, s0, c0);
        const float ra = rr ? a.wev_mfp[1][0] : a.wev_mfp[0][0], r2 = rr ? a.wev_mfp[1][1] : a.wev_mfp[0][1],
                    r3 = rr ? a.wev_mfp[1][2] : a.wev_mfp[0][2], et = rr ? a.wev_mfp[1][3] : a.wev_mfp[0][3];
        const float xx = fmaf(2.f * x0, d, x0 * x0);
        float f = fmaf(xx, r2, fmaf(2.f, ra, 1.f)) * ra * cs;
        f = fmaf(fmaf(xx, r2 - r3, ra + 1.f) * sn, __builtin_amdgcn_rcpf(x0 + d), f);
        f = fmaf(f, et, -2.f * r2);
        const float dd = fmaf(xx, r2, 1.f);
        return f * (-3.f * ra * __builtin_amdgcn_rcpf(dd * dd));
    };
    auto weval_one = [&](const KAbs &k, float R, int type, const float *tab, int rr) -> float {
        const float ph = __fmul_rn(k.kh, R);
        const float e = __fmaf_rn(k.kl, R, __fmaf_rn(k.kh, R, -ph));
        const float x0 = __fadd_rn(ph, e);
        if (type == 1) return ((double)x0 * 0.413566994 > 1) ? 0.f : 1.f;
        const float u = x0 * 4.0f;  // nodes at multiples of 1/4: exact
        if constexpr (WDIRECT)
            if (u >= (float)(a.wev_n_nodes - 2))
                return weval_direct(x0, __fadd_rn(__fsub_rn(ph, x0), e), type, rr);
        const int n = WDIRECT ? (int)u : min((int)u, a.wev_n_nodes - 2);
        const float t = u - (float)n;
        const float *nd = tab + 3 * n;
        const float f0 = nd[0], g0 = nd[1], q0 = nd[2], f1 = nd[3], g1 = nd[4], q1 = nd[5];
        const float A = ((f1 - f0) - g0) - q0, B = (g1 - g0) - 2.f * q0, C = q1 - q0;
        const float a3 = 10.f * A - 4.f * B + C, a4 = -15.f * A + 7.f * B - 2.f * C,
                    a5 = 6.f * A - 3.f * B + C;
        float p = fmaf(fmaf(fmaf(fmaf(fmaf(a5, t, a4), t, a3), t, q0), t, g0), t, f0);
        if (type == 3) {
            const float d = __fadd_rn(__fsub_rn(ph, x0), e);
            const float dp = fmaf(fmaf(fmaf(fmaf(5.f * a5, t, 4.f * a4), t, 3.f * a3), t, 2.f * q0), t, g0);
            p = fmaf(d * 4.0f, dp, p);
        }
        return p;
    };
    // k_x of this thread's rows (fixed for the whole kernel) -- float((double) index * dk), k_of()
    float wev_kx[WEVAL ? NW : 1];
    float wev_kx_half = 0.f;
    if constexpr (WEVAL) {
#pragma unroll
        for (int u = 0; u < NW; u++) {
            const int row = r0 + RSTEP * u;  // F512: u runs over all eight rows, |k_x| index = min(row, N - row)
            wev_kx[u] = (float)((double)(FUSED ? min(row, N - row) : row) * a.wev_dkx);
        }
        wev_kx_half = (float)((double)(N / 2) * a.wev_dkx);
    }
    auto eval_windows = [&](const LineItem &it, int m) {
        const int wsel = (a.dual && member_grid(it, m)) ? 1 : 0;
        const int type = wsel ? a.wev_type[1] : a.wev_type[0];
        const int per = 3 * a.wev_n_nodes;
        float kyc[2], kzc[2];  // the two columns of this thread
        if (it.filter_axis == 0) {
            kyc[0] = kyc[1] = (float)((double)it.og * a.wev_dky);  // og <= ny/2
            const int l0 = it.ct * TZ + 2 * c4;
            kzc[0] = (float)((double)l0 * a.wev_dkz);
            kzc[1] = (float)((double)(l0 + 1) * a.wev_dkz);
        } else {
            const int c0 = it.ct * TZ + 2 * c4;
            kyc[0] = (float)((double)min(c0, a.n_y - c0) * a.wev_dky);
            kyc[1] = (float)((double)min(c0 + 1, a.n_y - c0 - 1) * a.wev_dky);
            kzc[0] = kzc[1] = (float)((double)(a.n_z / 2) * a.wev_dkz);
        }
        const float *tab0 = wnodes + max(wsel ? a.wev_tab[0][1] : a.wev_tab[0][0], 0) * per;
        const float *tab1 = wnodes + max(wsel ? a.wev_tab[1][1] : a.wev_tab[1][0], 0) * per;
#pragma unroll
        for (int u = 0; u < NW; u++) {
            // F512: rows r0 + 64 u, u >= 4, are the mirror rows of thread (64 - r0, c4): it evaluates
            // them (as its u' = 7 - u) and hands them over below; row N/2 stays with r0 = 0
            if (FUSED && u >= NP && !(u == NP && r0 == 0)) continue;
            const KAbs k0 = k_abs(wev_kx[WEVAL ? u : 0], kyc[0], kzc[0]);
            const KAbs k1 = k_abs(wev_kx[WEVAL ? u : 0], kyc[1], kzc[1]);
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                const float R = rr ? a.wev_R[1] : a.wev_R[0];
                const float *tab = rr ? tab1 : tab0;
                (WPRE ? wpre[rr][WPRE ? u : 0] : wcur[rr][u]) =
                    make_float2(weval_one(k0, R, type, tab, rr), weval_one(k1, R, type, tab, rr));
            }
        }
        if (!FUSED && r0 == 0) {  // row N/2 (the mirror partner of row 0) belongs to the threads with r0 = 0
            const KAbs k0 = k_abs(wev_kx_half, kyc[0], kzc[0]), k1 = k_abs(wev_kx_half, kyc[1], kzc[1]);
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                const float R = rr ? a.wev_R[1] : a.wev_R[0];
                const float *tab = rr ? tab1 : tab0;
                (WPRE ? wpre_half[rr] : wcur_half[rr]) =
                    make_float2(weval_one(k0, R, type, tab, rr), weval_one(k1, R, type, tab, rr));
            }
        }
        if constexpr (FUSED) {  // the mirror halves change hands through the first tile buffer (free here)
            float2 *const exch = tile;
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < NR; rr++)
#pragma unroll
                for (int u = 0; u < NP; u++) exch[((int)threadIdx.x * NP + u) * NR + rr] = wcur[rr][u];
            __syncthreads();
            const int src = (((RSTEP - r0) & (RSTEP - 1)) * CPAIR + c4) * NP;
#pragma unroll
            for (int rr = 0; rr < NR; rr++)
#pragma unroll
                for (int u = NP; u < NW; u++)
                    // row r0 + 64 u mirrors to (64 - r0) + 64 (7 - u); for r0 = 0 that is 64 (8 - u)
                    if (!(u == NP && r0 == 0))
                        wcur[rr][u] = exch[(src + (NW - (r0 == 0 ? 0 : 1) - u)) * NR + rr];
            __syncthreads();
        }
    };
    auto issue_wloads = [&](const LineItem &it, int m) {
        if constexpr (WEVAL) {
            eval_windows(it, m);
            return;
        }
        const bool second = a.dual && member_grid(it, m);
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const wtab_t *t = second ? it.wt1 : it.wt0;
            if constexpr (PAIR)
                if (rr) t = second ? it.wt1b : it.wt0b;
            if (it.filter_axis == 0) {
                const unsigned wc = (unsigned)(a.n_z / 2);
                const wtab_t *b = t + (long)it.og * (N / 2 + 1) * wc + (it.ct * TZ + 2 * c4);
                if constexpr (FUSED) {  // one table row per tile row: index min(row, N - row)
#pragma unroll
                    for (int u = 0; u < NW; u++) {
                        const int row = r0 + RSTEP * u;
                        wcur[rr][u] = *reinterpret_cast<const float2 *>(b + (unsigned)min(row, N - row) * wc);
                    }
                    continue;
                }
#pragma unroll
                for (int u = 0; u < NP; u++)
                    (WPRE ? wpre[rr][WPRE ? u : 0] : wcur[rr][u]) =
                        *reinterpret_cast<const float2 *>(b + (unsigned)(r0 + RSTEP * u) * wc);
                (WPRE ? wpre_half[rr] : wcur_half[rr]) =
                    *reinterpret_cast<const float2 *>(b + (unsigned)(N / 2) * wc);
            } else {
                const int c0 = it.ct * TZ + 2 * c4;
                const unsigned nyh = (unsigned)(a.n_y / 2 + 1);
                const unsigned j0 = (unsigned)min(c0, a.n_y - c0),
                               j1 = (unsigned)min(c0 + 1, a.n_y - c0 - 1);
                if constexpr (FUSED) {
#pragma unroll
                    for (int u = 0; u < NW; u++) {
                        const int row = r0 + RSTEP * u;
                        const wtab_t *r = t + (unsigned)min(row, N - row) * nyh;
                        wcur[rr][u] = make_float2(r[j0], r[j1]);
                    }
                    continue;
                }
#pragma unroll
                for (int u = 0; u < NP; u++) {
                    const wtab_t *r = t + (unsigned)(r0 + RSTEP * u) * nyh;
                    (WPRE ? wpre[rr][WPRE ? u : 0] : wcur[rr][u]) = make_float2(r[j0], r[j1]);
                }
                {
                    const wtab_t *r = t + (unsigned)(N / 2) * nyh;
                    (WPRE ? wpre_half[rr] : wcur_half[rr]) = make_float2(r[j0], r[j1]);
                }
            }
        }
    };
    auto issue_loads = [&](float4(&reg)[2 * NP], const LineItem &it, int m) {
        const float2 *lo = (member_grid(it, m) ? it.src1 : it.src0) + member_base(it, m);
        const float2 *hi = lo + (long)((N / 2) >> it.line_lb) * it.line_bstride;
#pragma unroll
        for (int u = 0; u < 2 * NP; u++) {
            if constexpr (FUSED) {  // rows r0 + 64 u, the upper half addressed from the row-N/2 base
                const int row = r0 + RSTEP * u;
                const unsigned roff = row_off(it, row < N / 2 ? row : row - N / 2);
                reg[u] = *reinterpret_cast<const float4 *>((row < N / 2 ? lo : hi) + (roff + 2u * c4));
                continue;
            }
            const int row_a = r0 + RSTEP * (u >> 1);
            // mirror row N - row_a = N/2 + (N/2 - row_a); row_a = 0 pairs with N/2 itself
            const unsigned roff = (u & 1) ? (row_a == 0 ? 0u : row_off(it, N / 2 - row_a))
                                          : row_off(it, row_a);
            const float2 *p = (u & 1) ? hi : lo;
            reg[u] = *reinterpret_cast<const float4 *>(p + (roff + 2u * c4));
        }
    };

    // the tile sequence of this workgroup: members of an item, then the next item
    struct TileRef {
        LineItem it;
        int m, work, valid;
    };
    auto next_tile = [&](const TileRef &t) {
        // past the end: valid = 0, but (it, m) keep naming the last tile, so loads issued for
        // such a reference stay in bounds
        TileRef n = t;
        if (!t.valid) return n;
        if (t.m + 1 < t.it.npair * a.n_grids) {
            n.m = t.m + 1;
        } else {
            n.work = t.work + (int)gridDim.x;
            n.valid = n.work < n_work;
            if (n.valid) {
                n.m = 0;
                n.it = decode(n.work);
            }
        }
        return n;
    };

    bool first = true;
    // One tile: registers (x window) -> LDS, refill the register set with tile `refill`,
    // transform, store.  `nxt` is the tile after `cur` (its window loads are issued here).
    auto process = [&](float4(&reg)[2 * NP], const TileRef &cur, const TileRef &nxt,
                       const TileRef &refill) {
        const LineItem &it = cur.it;
        const int m = cur.m;
        const long st_half = (long)((N / 2) >> it.line_lb) * it.line_bstride;
        if (!first) __syncthreads();  // the previous tile's LDS reads are done
        first = false;
        if (WIN && WPRE && w_reload(it, m)) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
#pragma unroll
                for (int u = 0; u < NP; u++) wcur[rr][u] = wpre[rr][WPRE ? u : 0];
                wcur_half[rr] = wpre_half[rr];
            }
        }
        double op_c0 = 1., op_c1 = 1.;  // FMODE 4: sign * k_y^ey * k_z^ez of this thread's columns
        if (FMODE == 4) {
            const int c0 = it.ct * TZ + 2 * c4;
            int kyi0, kyi1, kzi0, kzi1;
            if (it.filter_axis == 0) {  // columns = k_z (< nz/2), outer = k_y
                const int mi = it.npair == 2 ? (m & 1) : 0;
                const int outer = mi == 0 ? it.og : it.n_outer - it.og;
                kyi0 = kyi1 = (outer <= a.n_y / 2) ? outer : outer - a.n_y;
                kzi0 = c0;
                kzi1 = c0 + 1;
            } else {  // Nyquist plane: columns = k_y, k_z = nz/2
                kyi0 = (c0 <= a.n_y / 2) ? c0 : c0 - a.n_y;
                kyi1 = (c0 + 1 <= a.n_y / 2) ? c0 + 1 : c0 + 1 - a.n_y;
                kzi0 = kzi1 = a.n_z / 2;
            }
            const double ky0 = (double)kyi0 * a.op_dky, ky1 = (double)kyi1 * a.op_dky;
            const double kz0 = (double)kzi0 * a.op_dkz, kz1 = (double)kzi1 * a.op_dkz;
            const double fy0 = a.op_ey == 0 ? 1. : (a.op_ey == 1 ? ky0 : ky0 * ky0);
            const double fy1 = a.op_ey == 0 ? 1. : (a.op_ey == 1 ? ky1 : ky1 * ky1);
            const double fz0 = a.op_ez == 0 ? 1. : (a.op_ez == 1 ? kz0 : kz0 * kz0);
            const double fz1 = a.op_ez == 0 ? 1. : (a.op_ez == 1 ? kz1 : kz1 * kz1);
            op_c0 = a.op_sign * fy0 * fz0;
            op_c1 = a.op_sign * fy1 * fz1;
        }
        // (two radii per sweep: the tile goes to LDS twice, once under each radius' window)
#pragma unroll
      for (int rr = 0; rr < NR; rr++) {
        float2 *const tile_r = tile + rr * N * TZ;
        if constexpr (F1024) {
            // radix-2 split, then the first stage of the two 512-point halves (tw_half[r0 j]):
            // a -> rows 8 r0 + j, b -> rows 512 + 8 r0 + j
            float2 wj[8];
            wj[1] = tw_half[r0];
            wj[2] = tw_half[2 * r0];
            wj[4] = tw_half[4 * r0];
            wj[3] = cmul(wj[1], wj[2]);
            wj[5] = cmul(wj[1], wj[4]);
            wj[6] = cmul(wj[2], wj[4]);
            wj[7] = cmul(wj[3], wj[4]);
            if (SIGN > 0)
#pragma unroll
                for (int j = 1; j < 8; j++) wj[j].y = -wj[j].y;
            float2 a0[8], a1[8], b0[8], b1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                float4 lo4 = reg[u], hi4 = reg[u + 8];
                if (WIN) {
                    const float2 wl = wcur[rr][u], wh = wcur[rr][u + 8];
                    lo4.x = __fmul_rn(lo4.x, wl.x);
                    lo4.y = __fmul_rn(lo4.y, wl.x);
                    lo4.z = __fmul_rn(lo4.z, wl.y);
                    lo4.w = __fmul_rn(lo4.w, wl.y);
                    hi4.x = __fmul_rn(hi4.x, wh.x);
                    hi4.y = __fmul_rn(hi4.y, wh.x);
                    hi4.z = __fmul_rn(hi4.z, wh.y);
                    hi4.w = __fmul_rn(hi4.w, wh.y);
                }
                float2 w = tw[r0 + RSTEP * u];
                if (SIGN > 0) w.y = -w.y;
                const float2 x0 = make_float2(lo4.x, lo4.y), x1 = make_float2(hi4.x, hi4.y);
                const float2 y0 = make_float2(lo4.z, lo4.w), y1 = make_float2(hi4.z, hi4.w);
                a0[u] = cadd(x0, x1);
                b0[u] = cmul(csub(x0, x1), w);
                a1[u] = cadd(y0, y1);
                b1[u] = cmul(csub(y0, y1), w);
            }
            Dft<8, SIGN>::run(a0);
            Dft<8, SIGN>::run(a1);
            Dft<8, SIGN>::run(b0);
            Dft<8, SIGN>::run(b1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float2 p0 = a0[j], p1 = a1[j], q0 = b0[j], q1 = b1[j];
                if (j > 0) {
                    p0 = cmul(p0, wj[j]);
                    p1 = cmul(p1, wj[j]);
                    q0 = cmul(q0, wj[j]);
                    q1 = cmul(q1, wj[j]);
                }
                *reinterpret_cast<float4 *>(tile_r + (8 * r0 + j) * TZ + 2 * c4) = make_float4(p0.x, p0.y, p1.x, p1.y);
                *reinterpret_cast<float4 *>(tile_r + (N / 2 + 8 * r0 + j) * TZ + 2 * c4) =
                    make_float4(q0.x, q0.y, q1.x, q1.y);
            }
            continue;
        }
        if constexpr (F512) {
            // first Stockham stage (s = 1, butterfly b = r0: inputs rows r0 + 64 k, outputs rows
            // 8 r0 + j times tw[r0 j]) on the registers, both columns
            float2 c0[8], c1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                float4 v = reg[u];
                if (WIN) {
                    const float2 wv = wcur[rr][u];
                    v.x = __fmul_rn(v.x, wv.x);
                    v.y = __fmul_rn(v.y, wv.x);
                    v.z = __fmul_rn(v.z, wv.y);
                    v.w = __fmul_rn(v.w, wv.y);
                }
                c0[u] = make_float2(v.x, v.y);
                c1[u] = make_float2(v.z, v.w);
            }
            if constexpr (PAIR && C21X_XPAIR_SKIP_FFT == 3) {
                // diagnostic (wrong results): the tile leaves as it came -- one read, two writes in this kernel's
                // access pattern, no LDS, no arithmetic: the data-movement ceiling of the two-radius pass X
                float2 *stp = member_grid(it, m) ? (rr ? it.dst1b : it.dst1) : (rr ? it.dst0b : it.dst0);
                stp += member_base(it, m);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int row = r0 + RSTEP * u;
                    const unsigned roff = row_off(it, row < N / 2 ? row : row - N / 2);
                    float2 *pp = stp + (row < N / 2 ? 0 : st_half);
                    *reinterpret_cast<float4 *>(pp + (roff + 2u * c4)) = reg[u];
                }
                continue;
            }
            if (!(PAIR && C21X_XPAIR_SKIP_FFT)) {
                Dft<8, SIGN>::run(c0);
                Dft<8, SIGN>::run(c1);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float2 o0 = c0[j], o1 = c1[j];
                if (j > 0 && !(PAIR && C21X_XPAIR_SKIP_FFT)) {
                    o0 = cmul(o0, twd1[F512_ ? j : 0]);
                    o1 = cmul(o1, twd1[F512_ ? j : 0]);
                }
                *reinterpret_cast<float4 *>(tile_r + (8 * r0 + j) * TZ + 2 * c4) =
                    make_float4(o0.x, o0.y, o1.x, o1.y);
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < 2 * NP; u++) {
            const int row_a = r0 + RSTEP * (u >> 1);
            const int row = (u & 1) ? mirror_row<N>(row_a) : row_a;
            const bool half = (u & 1) && row_a == 0;  // row N/2 has its own |k_x|
            float4 v = reg[u];
            if (FMODE == 4) {
                const int kxi = (row <= N / 2) ? row : row - N;
                const double kx = (double)kxi * a.op_dkx;
                const double fr = a.op_ex == 0 ? 1. : (a.op_ex == 1 ? kx : kx * kx);
                const double f0 = fr * op_c0, f1 = fr * op_c1;
                if (a.op_imag) {  // times i f
                    const float4 t = v;
                    v.x = (float)(-(double)t.y * f0);
                    v.y = (float)((double)t.x * f0);
                    v.z = (float)(-(double)t.w * f1);
                    v.w = (float)((double)t.z * f1);
                } else {
                    v.x = (float)((double)v.x * f0);
                    v.y = (float)((double)v.y * f0);
                    v.z = (float)((double)v.z * f1);
                    v.w = (float)((double)v.w * f1);
                }
            }
            if (WIN) {
                const float2 wv = half ? wcur_half[rr] : wcur[rr][u >> 1];
                v.x = __fmul_rn(v.x, wv.x);
                v.y = __fmul_rn(v.y, wv.x);
                v.z = __fmul_rn(v.z, wv.y);
                v.w = __fmul_rn(v.w, wv.y);
            }
            *reinterpret_cast<float4 *>(tile_r + row * TZ + 2 * c4) = v;
        }
      }
        __syncthreads();
        // ---- the register set is free: put the loads of the tile after next in flight
        // (unconditionally -- past the end of the sequence the last tile is simply read
        // again -- so that the number of loads in flight is static and the compiler can wait
        // with vmcnt(n > 0) for exactly this register set instead of draining everything)
        // (a TileRef past the end still names the last tile, see next_tile)
        if (WIN && !WEVAL && w_reload(nxt.it, nxt.m)) issue_wloads(nxt.it, nxt.m);
        issue_loads(reg, refill.it, refill.m);
        // (evaluated windows: ALU + LDS work only, placed behind the loads it can hide; F512: at the
        // end of the tile, where the first tile buffer is free for the exchange of the mirror halves)
        if (WEVAL && !FUSED && w_reload(nxt.it, nxt.m)) issue_wloads(nxt.it, nxt.m);

        if constexpr (F1024) {
            // second stage (s = 8) of both 512-point halves of the tile between one pair of barriers
            const int obase = (r0 & 7) + 64 * (r0 >> 3);
            float2 s0[2][8], s1[2][8];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float4 t = *reinterpret_cast<const float4 *>(tile + (h * (N / 2) + r0 + RSTEP * k) * TZ + 2 * c4);
                    s0[h][k] = make_float2(t.x, t.y);
                    s1[h][k] = make_float2(t.z, t.w);
                }
            __syncthreads();
            float2 w2[8];
            {
                const int ps = r0 & ~7;
                w2[1] = tw_half[ps];
                w2[2] = tw_half[2 * ps];
                w2[4] = tw_half[4 * ps];
                w2[3] = cmul(w2[1], w2[2]);
                w2[5] = cmul(w2[1], w2[4]);
                w2[6] = cmul(w2[2], w2[4]);
                w2[7] = cmul(w2[3], w2[4]);
                if (SIGN > 0)
#pragma unroll
                    for (int j = 1; j < 8; j++) w2[j].y = -w2[j].y;
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                Dft<8, SIGN>::run(s0[h]);
                Dft<8, SIGN>::run(s1[h]);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float2 o0 = s0[h][j], o1 = s1[h][j];
                    if (j > 0) {
                        o0 = cmul(o0, w2[j]);
                        o1 = cmul(o1, w2[j]);
                    }
                    *reinterpret_cast<float4 *>(tile + (h * (N / 2) + obase + 8 * j) * TZ + 2 * c4) =
                        make_float4(o0.x, o0.y, o1.x, o1.y);
                }
            }
            __syncthreads();
        }
        if constexpr (F512 && !(PAIR && C21X_XPAIR_SKIP_FFT)) {
            // second Stockham stage (s = 8) of all tiles of the sweep between ONE pair of barriers:
            // butterfly r0 of columns 2 c4, 2 c4 + 1: inputs rows r0 + 64 k, outputs rows
            // (r0 & 7) + 64 (r0 >> 3) + 8 j times tw[(r0 & ~7) j]
            constexpr int GRP = C21X_F512_MERGE ? NR : 1;  // tiles per barrier pair
            const int obase = (r0 & 7) + 64 * (r0 >> 3);
#pragma unroll
            for (int g = 0; g < NR; g += GRP) {
                float2 s0[GRP][8], s1[GRP][8];
#pragma unroll
                for (int rr = 0; rr < GRP; rr++)
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float4 t = *reinterpret_cast<const float4 *>(tile + (g + rr) * N * TZ +
                                                                           (r0 + RSTEP * k) * TZ + 2 * c4);
                        s0[rr][k] = make_float2(t.x, t.y);
                        s1[rr][k] = make_float2(t.z, t.w);
                    }
                __syncthreads();
                float2 w2[8];
                if (!C21X_F512_HOIST2) {
                    const int ps = r0 & ~7;
                    w2[1] = tw[ps];
                    w2[2] = tw[2 * ps];
                    w2[4] = tw[4 * ps];
                    w2[3] = cmul(w2[1], w2[2]);
                    w2[5] = cmul(w2[1], w2[4]);
                    w2[6] = cmul(w2[2], w2[4]);
                    w2[7] = cmul(w2[3], w2[4]);
                    if (SIGN > 0)
#pragma unroll
                        for (int j = 1; j < 8; j++) w2[j].y = -w2[j].y;
                }
#pragma unroll
                for (int rr = 0; rr < GRP; rr++) {
                    Dft<8, SIGN>::run(s0[rr]);
                    Dft<8, SIGN>::run(s1[rr]);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        float2 o0 = s0[rr][j], o1 = s1[rr][j];
                        if (j > 0) {
                            const float2 w = C21X_F512_HOIST2 ? twd2[F512_ ? j : 0] : w2[j];
                            o0 = cmul(o0, w);
                            o1 = cmul(o1, w);
                        }
                        *reinterpret_cast<float4 *>(tile + (g + rr) * N * TZ + (obase + 8 * j) * TZ + 2 * c4) =
                            make_float4(o0.x, o0.y, o1.x, o1.y);
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
      for (int rr = 0; rr < NR; rr++) {
        float2 *const tile_r = tile + rr * N * TZ;
        // store target of the tile (radius rr of the sweep)
        float2 *st_lo = member_grid(it, m) ? it.dst1 : it.dst0;
        if constexpr (PAIR)
            if (rr) st_lo = member_grid(it, m) ? it.dst1b : it.dst0b;
        st_lo += member_base(it, m);
        if constexpr (F1024) {
            // third stage of both halves on the values read back; A[m] = X[2 m], B[m] = X[2 m + 1],
            // m = r0 + 64 j: adjacent rows 2 m, 2 m + 1 of the line
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float2 c0[8], c1[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float4 t = *reinterpret_cast<const float4 *>(tile_r + (h * (N / 2) + r0 + RSTEP * k) * TZ + 2 * c4);
                    c0[k] = make_float2(t.x, t.y);
                    c1[k] = make_float2(t.z, t.w);
                }
                Dft<8, SIGN>::run(c0);
                Dft<8, SIGN>::run(c1);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float4 v = make_float4(c0[j].x, c0[j].y, c1[j].x, c1[j].y);
                    if (a.out_scale != 1.0f) {
                        v.x *= a.out_scale;
                        v.y *= a.out_scale;
                        v.z *= a.out_scale;
                        v.w *= a.out_scale;
                    }
                    const int row = 2 * (r0 + RSTEP * j) + h;
                    const unsigned roff = row_off(it, row < N / 2 ? row : row - N / 2);
                    float2 *p = st_lo + (row < N / 2 ? 0 : st_half);
                    *reinterpret_cast<float4 *>(p + (roff + 2u * c4)) = v;
                }
            }
            continue;
        }
        if constexpr (F512) {
            if constexpr (PAIR && C21X_XPAIR_SKIP_FFT == 3) continue;
            // third stage (s = 64, no twiddles: inputs rows r0 + 64 k, outputs rows r0 + 64 j) on the
            // values read back for the store
            float2 c0[8], c1[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float4 t = *reinterpret_cast<const float4 *>(tile_r + (r0 + RSTEP * k) * TZ + 2 * c4);
                c0[k] = make_float2(t.x, t.y);
                c1[k] = make_float2(t.z, t.w);
            }
            if (!(PAIR && C21X_XPAIR_SKIP_FFT)) {
                Dft<8, SIGN>::run(c0);
                Dft<8, SIGN>::run(c1);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float4 v = make_float4(c0[j].x, c0[j].y, c1[j].x, c1[j].y);
                if (a.out_scale != 1.0f) {
                    v.x *= a.out_scale;
                    v.y *= a.out_scale;
                    v.z *= a.out_scale;
                    v.w *= a.out_scale;
                }
                const int row = r0 + RSTEP * j;
                const unsigned roff = row_off(it, row < N / 2 ? row : row - N / 2);
                float2 *p = st_lo + (row < N / 2 ? 0 : st_half);
                *reinterpret_cast<float4 *>(p + (roff + 2u * c4)) = v;
            }
            continue;
        }
        if (!(PAIR && C21X_XPAIR_SKIP_FFT)) line_fft<N, TZ, SIGN, kBlock>(tile_r, tw, tw_half);
        if (PAIR && C21X_XPAIR_SKIP_FFT == 2 && rr == 1) continue;
        // ---- store
#pragma unroll
        for (int u = 0; u < 2 * NP; u++) {
            const int row_a = r0 + RSTEP * (u >> 1);
            const int row = (u & 1) ? mirror_row<N>(row_a) : row_a;
            float4 v = *reinterpret_cast<const float4 *>(tile_r + fft_out_row<N>(row) * TZ + 2 * c4);
            if (a.out_scale != 1.0f) {
                v.x *= a.out_scale;
                v.y *= a.out_scale;
                v.z *= a.out_scale;
                v.w *= a.out_scale;
            }
            const unsigned roff = (u & 1) ? (row_a == 0 ? 0u : row_off(it, N / 2 - row_a))
                                          : row_off(it, row_a);
            float2 *p = st_lo + ((u & 1) ? st_half : 0);
            *reinterpret_cast<float4 *>(p + (roff + 2u * c4)) = v;
        }
      }
        if (WEVAL && FUSED && w_reload(nxt.it, nxt.m)) issue_wloads(nxt.it, nxt.m);
    };

    if ((int)blockIdx.x < n_work) {
        TileRef ta;
        ta.work = blockIdx.x;
        ta.valid = 1;
        ta.m = 0;
        ta.it = decode(ta.work);
        if (WIN && !WEVAL) issue_wloads(ta.it, 0);
        issue_loads(reg_a, ta.it, 0);
        if (WEVAL) issue_wloads(ta.it, 0);
        TileRef tb = next_tile(ta);
        if (!TWO_SETS) {  // 1024-point lines, two radii per sweep: registers for one set only
            // The first tile is peeled here as well: at the loop head the operations in flight are
            // then the same on entry and on the back edge (this set's loads, then the stores of the
            // tile before), and the compiler waits for the loads alone.  Without it the merge of
            // "loads only" (entry) with "loads + stores" (back edge) made every tile start with
            // s_waitcnt vmcnt(0), i.e. with the drain of the previous tile's stores (round 3).
            process(reg_a, ta, tb, tb);
            while (tb.valid) {
                ta = tb;
                tb = next_tile(ta);
                process(reg_a, ta, tb, tb);
            }
            goto main_done;
        }
        issue_loads(reg_b, tb.it, tb.m);
        // The first tile is peeled so that at the loop head the memory operations in flight
        // are the same on entry and on the back edge (the other set's 8 loads + 8 stores): the
        // compiler then waits with the exact vmcnt for one register set, never for the stores.
        TileRef tc = next_tile(tb);
        process(reg_a, ta, tb, tc);
        while (tb.valid) {
            const TileRef td = next_tile(tc);
            process(reg_b, tb, tc, td);
            if (!tc.valid) break;
            const TileRef te = next_tile(td);
            process(reg_a, tc, td, te);
            tb = td;
            tc = te;
        }
    }
main_done:
    if (!(a.n_geo > 1 && a.g1_strided)) return;
    // ---- epilogue: the strided tiles of g1 (no window, not pipelined: 2 * nx/16 tiles in
    // all), taken by the workgroups at the end of the grid, which have the fewest main items
    for (int w = (int)(gridDim.x - 1 - blockIdx.x); w < n_work1; w += (int)gridDim.x) {
        // (the strided geometry is the unblocked Nyquist plane: offsets are i * bstride)
        const unsigned ls = (unsigned)a.g1.line_bstride, cs = (unsigned)a.g1.col_stride;
        const int nct = a.g1.n_ctiles;
        const int og = w / nct, ct = w - og * nct;
        const long base = (long)og * a.g1.outer_bstride + (long)ct * TZ * a.g1.col_stride;
        for (int g = 0; g < a.n_grids; g++) {
            const float2 *src = (g ? a.g1.src[1] : a.g1.src[0]) + base;
            float2 *dst = (g ? a.g1.dst[1] : a.g1.dst[0]) + base;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2 * NP; u++) {
                const int row_a = r0 + RSTEP * (u >> 1);
                const int row = (u & 1) ? mirror_row<N>(row_a) : row_a;
                const float2 e0 = src[(unsigned)row * ls + (unsigned)(2 * c4) * cs];
                const float2 e1 = src[(unsigned)row * ls + (unsigned)(2 * c4 + 1) * cs];
                *reinterpret_cast<float4 *>(tile + row * TZ + 2 * c4) =
                    make_float4(e0.x, e0.y, e1.x, e1.y);
            }
            __syncthreads();
            line_fft<N, TZ, SIGN, kBlock>(tile, tw, tw_half);
#pragma unroll
            for (int u = 0; u < 2 * NP; u++) {
                const int row_a = r0 + RSTEP * (u >> 1);
                const int row = (u & 1) ? mirror_row<N>(row_a) : row_a;
                float4 v = *reinterpret_cast<const float4 *>(tile + fft_out_row<N>(row) * TZ + 2 * c4);
                if (a.out_scale != 1.0f) {
                    v.x *= a.out_scale;
                    v.y *= a.out_scale;
                    v.z *= a.out_scale;
                    v.w *= a.out_scale;
                }
                dst[(unsigned)row * ls + (unsigned)(2 * c4) * cs] = make_float2(v.x, v.y);
                dst[(unsigned)row * ls + (unsigned)(2 * c4 + 1) * cs] = make_float2(v.z, v.w);
            }
        }
    }
}

// ------------------------------------------------------------------ pass Z (complex -> real)
// A real line of NZ points from its NZ/2+1 Hermitian coefficients via ONE complex FFT of
// length H = NZ/2:
//   E[k] = X[k] + conj(X[H-k]),  O[k] = (X[k] - conj(X[H-k])) * exp(+2 pi i k / NZ)
//   Z[k] = E[k] + i O[k]   ->   z = FFT_H^{+}(Z),   x[2j] = Re z[j],  x[2j+1] = Im z[j]
// (imaginary parts of X[0] and X[H] are ignored, as any c2r transform does).
// LZ = 16 consecutive lines per workgroup; LDS tile[k][line] with a row of LZ+1 so that both
// the transposing fill (lanes along k) and the FFT (lanes along line) are conflict free.
constexpr int LZ_PLAIN = 16;  // lines per workgroup, plain pass Z
constexpr int LZ_FUSED = 8;   // fused pass Z: smaller blocks -> 8 workgroups per CU

struct ZPassArgs {
    const float2 *main;  // [lines][H]
    const float2 *nyq;   // [lines]
    float *out;          // real rows of out_zstride floats
    long out_zstride;
    float out_scale;
    float out_div;       // != 0: stored value = v / out_div (EPI 0; the gathers of the ICs)
    int out_floor;       // != 0 (EPI 0): stored value >= -1 + 1e-7 (PerturbedField.c:262-264)
    int ny, lb;          // x-blocked layout: memory line -> logical line (logical_line())
    int reverse;         // wave-level kernel: workgroups walk the lines from the end
    // epilogues of the Eulerian source models (EPI 1, 2)
    double *p0, *p1;     // per-workgroup partials: EPI 1 min / max, EPI 2 sum (p0)
    float *f_out;        // EPI 2: dense f_coll grid [lines][NZ]
    double sig, delta_c;  // EPI 2: FgtrM_bias_fast (hmf.c:1205-1241); sig holds
                          // 1 / (growthf sqrt(2) sigma), < 0 when the sigmas coincide
    // EPI 3 (fill_Rbox_table / one_annular_filter, SpinTemperatureBox.c:606-629,713-731):
    // out = max(v, min_value) * const_factor, partials p0 min / p1 max / p2 sum of out
    double *p2;
    double min_value, const_factor;
    // EPI 4 (fused recombination loop): v = filtered whalo_sfr; where mask == r_index the dense
    // grid `out` holds delta_R of the crossing and receives
    // Gamma_12 = const_factor / (1 + delta_R) * max(v, 0)   (IonisationBox.c:1124-1140)
    const unsigned char *mask;
    int r_index;
    // EPI 5 (Eulerian source models with an x_e grid): v = filtered x_e of the cell, f_out holds the
    // radius' dense f_coll grid, *mean_dev its mean; the barrier of eulerian_mask_kernel
    // (IonisationBox.c:1091-1118) goes straight into the first-crossing mask -- x_e(R) is never stored
    unsigned char *mask_rw;
    const double *mean_dev;
    double mean_f_coll, f_limit, ion_eff;
    int fix_mean, mass_dep_zeta;
    // EPI 6 (closed-form Eulerian loop): EPI 2 for THIS radius plus the barrier of the PREVIOUS radius
    // of the loop -- its dense f_coll grid `f_prev`, its mean *mean_dev, its index r_index -- into
    // mask_rw.  A radius' barrier needs the box mean of its f_coll grid (IonisationBox.c:1022-1027), i.e.
    // a second sweep; riding the next radius' pass Z, which is bound by its butterflies and the erfc,
    // that sweep's 6 N bytes cost no time of their own (eulerian_mask_kernel: 0.15 ms per radius).
    const float *f_prev;
    // EPI 7 (closed-form Eulerian loop, banded barrier): EPI 2 for THIS radius whose barrier is decided in
    // the same sweep wherever it does not depend on the exact box mean: the correction mean_f_coll / mean
    // is known to lie in a band (predicted from the means of the radii before it, eul_band_kernel), the
    // barrier is monotone in it and in the cell's f_coll, so each end of the band is a threshold on the
    // float f_coll: band[0] (at or above: crosses for sure), band[1] (below: does not).  The cells in
    // between get the marker 255 in mask_rw and their f_coll in f_out (a sparse write: f_out is NOT a
    // complete grid), and are settled with the exact threshold by the next radius' sweep (r_prev >= 0: the
    // radius whose markers are outstanding, *mean_dev ITS EXACT THRESHOLD) or by eul_resolve_pending_kernel.
    const double *band;
    int r_prev;
    // EPI 8 (Eulerian table models with an x_e grid, banded barrier): EPI 5 and fcoll_eulerian_kernel in
    // ONE sweep.  v = filtered x_e of the cell; delta_fil = the radius' dense (padded, out_zstride floats
    // per row) filtered density, table / tab_min / tab_width / tab_mode its f_coll table; the f_coll sum
    // goes to p0 like EPI 2; the barrier f mf zeta > 1 - x_e is monotone in the mean fix mf, which is
    // known to lie in [band[0], band[1]] (eul_band_step, mf space): cells both ends agree on are final,
    // the others get the marker 255 and leave (f, clipped x_e) in f_out / xe_pend (sparse, dense layout)
    // for the next radius' sweep (r_prev >= 0, *mean_dev = ITS EXACT mean fix) or
    // eul_resolve_pending_xe_kernel.  The dense f_coll grid is neither written nor read.
    const float *delta_fil;
    const float *table;
    float *xe_pend;
    double tab_min, tab_width;
    int tab_mode;
};

// --- building blocks shared by the plain and the fused pass-Z kernels
template <int NZ, int LZ>
struct ZGeom {
    static constexpr int H = NZ / 2;
    static constexpr int ZROW = LZ + 1;
    static constexpr int NF4 = LZ * H / 2;                       // float4 per 16-line block
    static constexpr int NLOAD = (NF4 + kBlock - 1) / kBlock;    // float4 loads per thread
    static constexpr int NOUT = LZ * H / kBlock;                 // float2 outputs per thread
};

template <int NZ, int LZ>
__device__ __forceinline__ void z_issue_loads(const float2 *main, long l0,
                                              float4 (&reg)[ZGeom<NZ, LZ>::NLOAD]) {
    constexpr int H = ZGeom<NZ, LZ>::H, NF4 = ZGeom<NZ, LZ>::NF4;
    const float4 *src4 = reinterpret_cast<const float4 *>(main + l0 * H);
#pragma unroll
    for (int u = 0; u < ZGeom<NZ, LZ>::NLOAD; u++) {
        const int f = threadIdx.x + kBlock * u;
        if (NF4 % kBlock == 0 || f < NF4) reg[u] = src4[f];
    }
}

// registers -> tile[k][line] (transposing), Hermitian pre-processing, length-H inverse FFT
template <int NZ, int LZ>
__device__ __forceinline__ void z_transform(float2 *tile, const float2 *twH, const float2 *twN,
                                            const float4 (&reg)[ZGeom<NZ, LZ>::NLOAD],
                                            const float2 *nyq, long l0, int ny, int lb) {
    constexpr int H = ZGeom<NZ, LZ>::H, NF4 = ZGeom<NZ, LZ>::NF4, ZROW = LZ + 1;
#pragma unroll
    for (int u = 0; u < ZGeom<NZ, LZ>::NLOAD; u++) {
        const int f = threadIdx.x + kBlock * u;
        if (NF4 % kBlock == 0 || f < NF4) {
            const int e = 2 * f;
            const int li = e / H, k = e % H;
            tile[k * ZROW + li] = make_float2(reg[u].x, reg[u].y);
            tile[(k + 1) * ZROW + li] = make_float2(reg[u].z, reg[u].w);
        }
    }
    __syncthreads();
    // pairs (k, H-k) are owned by one thread
    constexpr int NPRE = (H / 2 + 1) * LZ;
    for (int i = threadIdx.x; i < NPRE; i += kBlock) {
        const int li = i % LZ, k = i / LZ;
        if (k == 0) {
            const float x0 = tile[li].x;
            const float xh = nyq[logical_line(l0 + li, ny, lb)].x;
            tile[li] = make_float2(x0 + xh, x0 - xh);
        } else {
            const float2 A = tile[k * ZROW + li], B = tile[(H - k) * ZROW + li];
            const float2 E = make_float2(A.x + B.x, A.y - B.y);  // A + conj(B)
            const float2 D = make_float2(A.x - B.x, A.y + B.y);  // A - conj(B)
            float2 w = twN[k];
            w.y = -w.y;  // exp(+2 pi i k / NZ)
            const float2 O = cmul(D, w);
            tile[k * ZROW + li] = make_float2(E.x - O.y, E.y + O.x);
            tile[(H - k) * ZROW + li] = make_float2(E.x + O.y, O.x - E.y);
        }
    }
    __syncthreads();
    fft_tile<H, LZ, ZROW, +1, kBlock>(tile, twH);
}

// EPI 0: store the real lines.  EPI 1: store them and emit the workgroup's min / max (the
// extrema the host needs for the per-radius f_coll table, IonisationBox.c:668-699).  EPI 2:
// CONST-ION-EFF closed form: f_coll(delta_R) per cell straight from the tile (clips of
// :689,803, FgtrM_bias_fast) to the dense grid + the workgroup's partial of its sum (:785-961);
// the filtered density itself is never written.  EPI 3: the floor-and-scale store of the
// spin-temperature filter tables with min / max / sum of the stored values.
template <int NZ, int EPI>
__global__ void __launch_bounds__(kBlock)
z_c2r_kernel(ZPassArgs a, const float2 *__restrict__ twH_global,
             const float2 *__restrict__ twN_global) {
    constexpr int H = NZ / 2, LZ = LZ_PLAIN, ZROW = LZ + 1;
    extern __shared__ float4 lds_raw[];
    float2 *tile = reinterpret_cast<float2 *>(lds_raw);  // [H][ZROW]
    float2 *twH = tile + H * ZROW;                       // [H]
    float2 *twN = twH + H;                               // [H/2 + 1]
    for (int t = threadIdx.x; t < H; t += kBlock) twH[t] = twH_global[t];
    for (int t = threadIdx.x; t <= H / 2; t += kBlock) twN[t] = twN_global[t];

    const long l0 = (long)blockIdx.x * LZ;
    float4 reg[ZGeom<NZ, LZ>::NLOAD];
    z_issue_loads<NZ, LZ>(a.main, l0, reg);
    z_transform<NZ, LZ>(tile, twH, twN, reg, a.nyq, l0, a.ny, a.lb);
    // ---- store: lanes along j, one float2 = (x[2j], x[2j+1])
    double acc0 = 0., acc1 = 0., acc2 = 0.;
#pragma unroll
    for (int u = 0; u < ZGeom<NZ, LZ>::NOUT; u++) {
        const int f = threadIdx.x + kBlock * u;
        const int li = f / H, j = f % H;
        float2 v = tile[j * ZROW + li];
        if (a.out_scale != 1.0f) {
            v.x *= a.out_scale;
            v.y *= a.out_scale;
        }
        if (EPI == 0 && a.out_div != 0.f) {
            v.x = __fdiv_rn(v.x, a.out_div);
            v.y = __fdiv_rn(v.y, a.out_div);
        }
        if (EPI == 0 && a.out_floor) {
            if ((double)v.x < -1.0 + 1e-7) v.x = (float)(-1.0 + 1e-7);
            if ((double)v.y < -1.0 + 1e-7) v.y = (float)(-1.0 + 1e-7);
        }
        if (EPI == 3) {
            // float compared with the double floor, float x double product rounded to float
            if ((double)v.x < a.min_value) v.x = (float)a.min_value;
            if ((double)v.y < a.min_value) v.y = (float)a.min_value;
            v.x = (float)((double)v.x * a.const_factor);
            v.y = (float)((double)v.y * a.const_factor);
            acc2 += (double)v.x;
            acc2 += (double)v.y;
        }
        if (EPI == 2) {
            const double f0 = (a.sig < 0) ? 0. : fgtrm_bias_fast_inv(clip_delta_eulerian(v.x), a.sig, a.delta_c);
            const double f1 = (a.sig < 0) ? 0. : fgtrm_bias_fast_inv(clip_delta_eulerian(v.y), a.sig, a.delta_c);
            acc0 += f0;
            acc0 += f1;
            reinterpret_cast<float2 *>(a.f_out + logical_line(l0 + li, a.ny, a.lb) * NZ)[j] =
                make_float2((float)f0, (float)f1);
        } else {
            reinterpret_cast<float2 *>(a.out + logical_line(l0 + li, a.ny, a.lb) * a.out_zstride)[j] = v;
            if (EPI == 1 || EPI == 3) {
                const double lo = fmin((double)v.x, (double)v.y), hi = fmax((double)v.x, (double)v.y);
                acc0 = (u == 0) ? lo : fmin(acc0, lo);
                acc1 = (u == 0) ? hi : fmax(acc1, hi);
            }
        }
    }
    if (EPI != 0) {
        __shared__ double red0[kBlock / 64], red1[kBlock / 64], red2[kBlock / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o0 = __shfl_down(acc0, off, 64), o1 = __shfl_down(acc1, off, 64);
            acc0 = (EPI == 2 || EPI == 6 || EPI == 7 || EPI == 8) ? acc0 + o0 : fmin(acc0, o0);
            acc1 = fmax(acc1, o1);
            if (EPI == 3) acc2 += __shfl_down(acc2, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red0[threadIdx.x >> 6] = acc0;
            red1[threadIdx.x >> 6] = acc1;
            red2[threadIdx.x >> 6] = acc2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double r0 = red0[0], r1 = red1[0], r2 = red2[0];
#pragma unroll
            for (int w = 1; w < kBlock / 64; w++) {
                r0 = (EPI == 2 || EPI == 6 || EPI == 7 || EPI == 8) ? r0 + red0[w] : fmin(r0, red0[w]);
                r1 = fmax(r1, red1[w]);
                r2 += red2[w];
            }
            a.p0[blockIdx.x] = r0;
            if (EPI == 1 || EPI == 3) a.p1[blockIdx.x] = r1;
            if (EPI == 3) a.p2[blockIdx.x] = r2;
        }
    }
}

// ------------------------------------------------------------------ pass Z (real -> complex)
// Forward counterpart: z[j] = x[2j] + i x[2j+1], Zf = FFT_H^{-}(z), then
//   X[k]   = E - i w_k D,   X[H-k] = conj(E) - i conj(w_k D),   w_k = exp(-2 pi i k / NZ),
//   E = (Zf[k] + conj Zf[H-k]) / 2,  D = (Zf[k] - conj Zf[H-k]) / 2,
//   X[0] = Re Zf[0] + Im Zf[0],  X[H] = Re Zf[0] - Im Zf[0]  (-> Nyquist plane).
// The dense input is read with the scale-and-clip of prepare_box_for_filtering
// (IonisationBox.c:333-350) applied on load, so no packing sweep is needed.
struct ZFwdArgs {
    const float *in;  // real rows of in_zstride floats
    long in_zstride;
    float2 *main, *nyq;
    double factor, lo, hi;  // v = clip(in*factor, lo, hi); clip disabled when lo > hi
    int ny, lb;             // x-blocked layout (logical_line())
};

template <int NZ>
__global__ void __launch_bounds__(kBlock)
z_r2c_kernel(ZFwdArgs a, const float2 *__restrict__ twH_global,
             const float2 *__restrict__ twN_global) {
    constexpr int H = NZ / 2, LZ = LZ_PLAIN, ZROW = LZ + 1;
    extern __shared__ float4 lds_raw[];
    float2 *tile = reinterpret_cast<float2 *>(lds_raw);  // [H][ZROW]
    float2 *twH = tile + H * ZROW;
    float2 *twN = twH + H;
    for (int t = threadIdx.x; t < H; t += kBlock) twH[t] = twH_global[t];
    for (int t = threadIdx.x; t <= H / 2; t += kBlock) twN[t] = twN_global[t];

    const long l0 = (long)blockIdx.x * LZ;
    const bool clip = a.lo <= a.hi;
    constexpr int NIN = LZ * H / kBlock;  // float2 per thread
#pragma unroll
    for (int u = 0; u < NIN; u++) {
        const int f = threadIdx.x + kBlock * u;
        const int li = f / H, j = f % H;
        float2 v = reinterpret_cast<const float2 *>(a.in + logical_line(l0 + li, a.ny, a.lb) * a.in_zstride)[j];
        if (clip) {
            v.x = (float)fmax(fmin((double)v.x * a.factor, a.hi), a.lo);
            v.y = (float)fmax(fmin((double)v.y * a.factor, a.hi), a.lo);
        } else if (a.factor != 1.0) {
            v.x = (float)((double)v.x * a.factor);
            v.y = (float)((double)v.y * a.factor);
        }
        tile[j * ZROW + li] = v;
    }
    __syncthreads();
    fft_tile<H, LZ, ZROW, -1, kBlock>(tile, twH);
    // Hermitian post-processing, pairs (k, H-k) owned by one thread
    constexpr int NPRE = (H / 2 + 1) * LZ;
    for (int i = threadIdx.x; i < NPRE; i += kBlock) {
        const int li = i % LZ, k = i / LZ;
        if (k == 0) {
            const float2 z0 = tile[li];
            tile[li] = make_float2(z0.x + z0.y, 0.f);
            a.nyq[logical_line(l0 + li, a.ny, a.lb)] = make_float2(z0.x - z0.y, 0.f);
        } else {
            const float2 A = tile[k * ZROW + li], B = tile[(H - k) * ZROW + li];
            const float2 E = make_float2(0.5f * (A.x + B.x), 0.5f * (A.y - B.y));
            const float2 D = make_float2(0.5f * (A.x - B.x), 0.5f * (A.y + B.y));
            const float2 wD = cmul(D, twN[k]);  // w_k D
            // X[k] = E - i wD ; X[H-k] = conj(E) - i conj(wD)
            tile[k * ZROW + li] = make_float2(E.x + wD.y, E.y - wD.x);
            tile[(H - k) * ZROW + li] = make_float2(E.x - wD.y, -E.y - wD.x);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NIN; u++) {
        const int f = threadIdx.x + kBlock * u;
        const int li = f / H, k = f % H;
        a.main[(l0 + li) * H + k] = tile[k * ZROW + li];
    }
}

// ------------------------------------------------------------------ fused pass Z + barrier
// Lagrangian source grids, radius index > 0: the z-lines of BOTH filtered grids (delta_R and
// the filtered emissivity) are transformed in one workgroup and consumed in registers:
//   sum(max(stars,0))                        calculate_fcoll_grid   IonisationBox.c:821-837,954
//   stars*zeta / (rho_b (1+delta_R)) > 1     find_ionised_regions   IonisationBox.c:1054-1082,1118
// The filtered real-space grids are never written to HBM.  The only output is the per-cell
// first-crossing radius index (uint8, read-modify-write of full rows) from which the
// driver derives xH = 0 / z_reion after the loop; because radii are visited largest first,
// "mask == 0 ? r : mask" records the first crossing exactly like the reference's in-loop
// xH / z_reion writes.  The barrier is evaluated division-free, stars*zeta > rho_b(1+delta)
// (and the f_limit floor as a uniform predicate), which is the same inequality in exact
// arithmetic and differs from the reference's rounding only for cells within 1 ulp (double)
// of the barrier.
struct ZFusedArgs {
    const float2 *d_main, *d_nyq;  // filtered density spectrum after passes X, Y
    const float2 *s_main, *s_nyq;  // filtered emissivity spectrum
    const float2 *x_main, *x_nyq;  // filtered x_e spectrum (USE_TS_FLUCT; wave-level kernel only)
    unsigned char *first_cross;    // [lines][NZ]
    double *partials;              // one per workgroup (nx*ny/LZ_FUSED)
    double rhocrit_omb, ion_eff, f_limit;
    int mass_dep_zeta, r_index;
    int ny, lb;  // x-blocked layout (logical_line())
    int store_all;  // write the mask back even where this radius changed nothing
    int reverse;    // workgroups walk the lines from the end (see dispatch_z_fused)
    // Recombination models with CELL_RECOMB (wave-level kernel, RC = true): the third spectrum
    // (x_main / x_nyq) is the filtered HaloBox.whalo_sfr; the barrier gains (1 + rec),
    // rec = N_rec / (1 + delta_R) with the cell's own previous N_rec (`nrec`, dense; NULL: the
    // homogeneous model's one number `rec0`), and a cell's FIRST crossing records
    // Gamma_12 = R gamma_prefactor / (1 + delta_R) max(sfr_R, 0)  (IonisationBox.c:1084-1140)
    int rc;
    const float *nrec;
    double rec0;
    // the third line (x_main) is not x_e but the previous snapshot's N_rec filtered at this radius
    // (CELL_RECOMB = false, IonisationBox.c:583-663,808-809,1093): rec = max(N_rec(R), 0) / (1 + delta_R)
    int x_is_nrec;
    // ... or BOTH: x_main is x_e and n_main the filtered N_rec, a fourth spectrum whose transform is
    // parked (clipped at zero) in the LDS rows that hold the dense N_rec otherwise (512-point lines)
    const float2 *n_main, *n_nyq;
    float *g12;  // dense [lines][NZ]: a first crossing leaves delta_R here (-> zw_c2r_kernel EPI 4)
};

template <int NZ>
__global__ void __launch_bounds__(kBlock)
z_c2r_ionise_kernel(ZFusedArgs a, const float2 *__restrict__ twH_global,
                    const float2 *__restrict__ twN_global) {
    constexpr int H = NZ / 2, LZ = LZ_FUSED, ZROW = LZ + 1;
    constexpr int NOUT = ZGeom<NZ, LZ>::NOUT;
    extern __shared__ float4 lds_raw[];
    float2 *tile = reinterpret_cast<float2 *>(lds_raw);
    float2 *twH = tile + H * ZROW;
    float2 *twN = twH + H;
    for (int t = threadIdx.x; t < H; t += kBlock) twH[t] = twH_global[t];
    for (int t = threadIdx.x; t <= H / 2; t += kBlock) twN[t] = twN_global[t];

    const long l0 = (long)blockIdx.x * LZ;
    float4 reg_d[ZGeom<NZ, LZ>::NLOAD], reg_s[ZGeom<NZ, LZ>::NLOAD];
    z_issue_loads<NZ, LZ>(a.d_main, l0, reg_d);
    z_issue_loads<NZ, LZ>(a.s_main, l0, reg_s);  // in flight during the first transform
    // the mask rows of this block, also early
    uchar2 old[NOUT];
#pragma unroll
    for (int u = 0; u < NOUT; u++) {
        const int f = threadIdx.x + kBlock * u;
        const int li = f / H, j = f % H;
        old[u] = reinterpret_cast<const uchar2 *>(a.first_cross + logical_line(l0 + li, a.ny, a.lb) * NZ)[j];
    }
    z_transform<NZ, LZ>(tile, twH, twN, reg_d, a.d_nyq, l0, a.ny, a.lb);
    float2 dens[NOUT];
#pragma unroll
    for (int u = 0; u < NOUT; u++) {
        const int f = threadIdx.x + kBlock * u;
        const int li = f / H, j = f % H;
        dens[u] = tile[j * ZROW + li];
    }
    __syncthreads();
    z_transform<NZ, LZ>(tile, twH, twN, reg_s, a.s_nyq, l0, a.ny, a.lb);

    const bool floor_ionises = a.mass_dep_zeta && (a.f_limit * a.ion_eff > 1.);
    const float dmin = (float)(-1. + 1e-7);  // IonisationBox.c:803
    double acc = 0.;
#pragma unroll
    for (int u = 0; u < NOUT; u++) {
        const int f = threadIdx.x + kBlock * u;
        const int li = f / H, j = f % H;
        const float2 st = tile[j * ZROW + li];
        const float s0 = fmaxf(st.x, 0.f), s1 = fmaxf(st.y, 0.f);
        acc += (double)s0;
        acc += (double)s1;
        const double D0 = a.rhocrit_omb * (1. + (double)fmaxf(dens[u].x, dmin));
        const double D1 = a.rhocrit_omb * (1. + (double)fmaxf(dens[u].y, dmin));
        const bool i0 = floor_ionises || ((double)s0 * a.ion_eff > D0);
        const bool i1 = floor_ionises || ((double)s1 * a.ion_eff > D1);
        uchar2 m = old[u];
        const bool n0 = i0 && m.x == 0, n1 = i1 && m.y == 0;
        if (n0) m.x = (unsigned char)a.r_index;
        if (n1) m.y = (unsigned char)a.r_index;
        if (n0 || n1 || a.store_all)
            reinterpret_cast<uchar2 *>(a.first_cross + logical_line(l0 + li, a.ny, a.lb) * NZ)[j] = m;
    }
    // workgroup partial of sum(stars)
    __shared__ double red[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) sum += red[w];
        a.partials[blockIdx.x] = sum;
    }
}

// ------------------------------------------------------------------ layout conversion
// FFTW-style padded half-spectrum [lines][H+1] -> split (main [lines][H], nyq [lines])
__global__ void __launch_bounds__(kBlock)
padded_to_split_kernel(const float2 *__restrict__ padded, float2 *__restrict__ main,
                       float2 *__restrict__ nyq, long nlines, int H, int ny, int lb) {
    const long total = nlines * (long)(H + 1);
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (long)gridDim.x * kBlock) {
        const long line = i / (H + 1);
        const int k = (int)(i - line * (H + 1));
        const float2 v = padded[i];
        if (k < H) {
            // logical line x * ny + y -> memory line (x / XB * ny + y) * XB + x % XB
            const long x = line / ny, y = line - x * ny;
            const long m = lb ? (((x >> lb) * ny + y) << lb) + (x & ((1 << lb) - 1)) : line;
            main[m * H + k] = v;
        } else {
            nyq[line] = v;
        }
    }
}

// ------------------------------------------------------------------ host side
struct Twiddles {
    float2 *dev = nullptr;
};
std::map<int, Twiddles> g_tw;
std::mutex g_tw_mutex;

// exp(-2 pi i t / n), t = 0..n-1, computed in double
const float2 *twiddles(int n) {
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto it = g_tw.find(n);
    if (it != g_tw.end()) return it->second.dev;
    std::vector<float2> host(n);
    for (int t = 0; t < n; t++) {
        const double ang = -2.0 * M_PI * (double)t / (double)n;
        host[t] = make_float2((float)cos(ang), (float)sin(ang));
    }
    Twiddles tw;
    if (hipMalloc(&tw.dev, sizeof(float2) * n) != hipSuccess) return nullptr;
    if (hipMemcpy(tw.dev, host.data(), sizeof(float2) * n, hipMemcpyHostToDevice) != hipSuccess)
        return nullptr;
    g_tw[n] = tw;
    return tw.dev;
}

bool pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// The four line geometries of the split layout; `main` points at the [nx][ny][H] block of a
// grid (its Nyquist plane follows at main + nx*ny*H).  src/dst of grid g go to slot g.
void geo_ptrs(LineGeo &g, int grid, const float2 *src, float2 *dst) {
    g.src[grid] = src;
    g.dst[grid] = dst;
}
// offsets are (i >> lb) * bstride + (i & (2^lb - 1)) * stride; unblocked axes use lb = 0
void geo_axes(LineGeo &g, int line_lb, long line_stride, long line_bstride, int outer_lb,
              long outer_stride, long outer_bstride) {
    g.line_lb = line_lb;
    g.line_stride = line_stride;
    g.line_bstride = line_bstride;
    g.outer_lb = outer_lb;
    g.outer_stride = outer_stride;
    g.outer_bstride = outer_bstride;
}
// main block [x / XB][y][x % XB][k_z], XB = 2^lb (lb = split_xb_log2(nx))
LineGeo geo_x_main(int ny, int H, int tz, int lb) {  // lines along x, outer = k_y (mirror-paired), columns = k_z
    LineGeo g{};
    const long xb = 1L << lb;
    geo_axes(g, lb, H, (long)ny * xb * H, 0, 0, xb * H);
    g.col_stride = 1;
    g.n_outer = ny;
    g.n_ctiles = H / tz;
    g.pair_outer = 1;
    g.filter_axis = 0;
    return g;
}
LineGeo geo_x_nyq(int ny, int tz) {  // Nyquist plane [nx][ny]: lines along x, columns = k_y
    LineGeo g{};
    geo_axes(g, 0, 0, ny, 0, 0, 0);
    g.col_stride = 1;
    g.n_outer = 1;
    g.n_ctiles = ny / tz;
    g.pair_outer = 0;
    g.filter_axis = 1;
    return g;
}
LineGeo geo_y_main(int nx, int ny, int H, int tz, int lb) {  // lines along y, outer = x, columns = k_z
    LineGeo g{};
    const long xb = 1L << lb;
    geo_axes(g, 0, 0, xb * H, lb, H, (long)ny * xb * H);
    g.col_stride = 1;
    g.n_outer = nx;
    g.n_ctiles = H / tz;
    return g;
}
LineGeo geo_y_nyq(int nx, int ny, int tz) {  // Nyquist plane: lines along y contiguous, columns = x
    LineGeo g{};
    geo_axes(g, 0, 0, 1, 0, 0, 0);
    g.col_stride = ny;
    g.n_outer = 1;
    g.n_ctiles = nx / tz;
    return g;
}

// ---- in-loop kernel timing (bench.py): with c21hip_ktime_enable(1) every pass launch is bracketed
// by two HIP events on ITS stream; c21hip_ktime_report sums them per kernel kind.  Kinds are those
// of c21hip_bench_pass: 0 pass X with streamed tables, 1 pass Y, 2 fused pass Z, 6 two-radius pass X
// with tables, 7 / 8 pass X / two-radius pass X with evaluated windows, 9 forward line passes.
struct KTimeRec {
    int kind;
    hipEvent_t e0, e1;
};
std::vector<KTimeRec> g_ktime;
int g_ktime_on = 0;
struct KTimeScope {
    KTimeRec r{-1, nullptr, nullptr};
    hipStream_t stream;
    KTimeScope(int kind, hipStream_t s) : stream(s) {
        if (!g_ktime_on) return;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        r.kind = kind;
        (void)hipEventRecord(r.e0, stream);
    }
    ~KTimeScope() {
        if (r.kind < 0) return;
        (void)hipEventRecord(r.e1, stream);
        g_ktime.push_back(r);
    }
};
constexpr int ktime_kind(int sign, int fmode) {
    return sign < 0 ? 9 : (fmode == 0 ? 1 : fmode == 3 ? 0 : fmode == 5 ? 6 : (fmode == 6 || fmode == 8) ? 7
                                                              : (fmode == 7 || fmode == 9) ? 8 : 10);
}

#ifndef C21X_XORDER_DEFAULT
#define C21X_XORDER_DEFAULT 3
#endif
template <int N, int SIGN, int FMODE>
int launch_line_pass_mode(const LinePassArgs &a, hipStream_t stream) {
    const float2 *tw = twiddles(N);
    if (!tw) {
        c21hip_set_error("native FFT: twiddle table allocation failed");
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    const size_t lds = sizeof(float2) * ((size_t)((FMODE == 5 || FMODE == 7 || FMODE == 9) ? 2 : 1) * N *
                                             line_tile_cols(N) + N + (N >= 1024 ? N / 2 : 0)) +
                       ((FMODE >= 6 && FMODE <= 9) ? sizeof(float) * 3 * (size_t)a.wev_n_nodes * a.wev_n_tabs : 0);
    if (lds > 160 * 1024) {
        c21hip_set_error("native FFT: %zu bytes of LDS for a %d-point line pass (mode %d)", lds, N, FMODE);
        return C21CM_VALUE_ERROR;
    }
    const int n_work = geo_items(a.g0) + (a.n_geo > 1 ? geo_items(a.g1) : 0);
    // persistent grid: as many workgroups as fit (LDS-limited), each striding over the work
    int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
    // lines of 256 points and fewer leave LDS and registers for a second workgroup per CU (2 %)
    const int by_waves = (LineThreads<N, FMODE>::value >= 512 && N > 256) ? 1 : 2;
    if (per_cu > by_waves) per_cu = by_waves;
    int nblocks = 256 * per_cu;
    if (nblocks > n_work) nblocks = n_work;
    static size_t attr_lds = 0;  // (the evaluated-window modes size their LDS by the node count)
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute((const void *)line_pass_kernel<N, SIGN, FMODE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    LinePassArgs ao = a;
    {
        // item order of the main geometry: rotate the column tile with the trip (default, round 5: a
        // workgroup no longer meets the same column tile on every trip -- pass Y 0.476 -> 0.447 ms at
        // 512^3, 3.45 -> 2.97 ms at 1024^3); C21CM_YORDER / C21CM_XORDER = 0 restore the plain order,
        // 1 / k select the experimental orders of DESIGN section 8
        static int xo = -99, yo = -99;
        if (xo == -99) {
            const char *ex = getenv("C21CM_XORDER"), *ey = getenv("C21CM_YORDER");
            xo = ex ? atoi(ex) : C21X_XORDER_DEFAULT;
            yo = ey ? atoi(ey) : -1;
        }
        ao.item_order = (FMODE == 0) ? yo : xo;
    }
    KTimeScope kt(ktime_kind(SIGN, FMODE), stream);
    hipLaunchKernelGGL((line_pass_kernel<N, SIGN, FMODE>), dim3((unsigned)nblocks),
                       dim3(LineThreads<N, FMODE>::value), lds, stream, ao, tw);
    LAUNCH_CHECK();
    return 0;
}

template <int N, int SIGN>
int launch_line_pass(const LinePassArgs &a, int fmode, hipStream_t stream) {
    if (SIGN > 0 && fmode == 3) return launch_line_pass_mode<N, +1, 3>(a, stream);
    if (SIGN > 0 && fmode == 4) return launch_line_pass_mode<N, +1, 4>(a, stream);
    if constexpr (N <= 512)  // two radii per sweep: two tiles in LDS
        if (SIGN > 0 && fmode == 5) return launch_line_pass_mode<N, +1, 5>(a, stream);
    if constexpr (N >= 128 && (N & (N - 1)) == 0) {  // windows evaluated in the kernel
        if (SIGN > 0 && fmode == 6) return launch_line_pass_mode<N, +1, 6>(a, stream);
        if (SIGN > 0 && fmode == 8) return launch_line_pass_mode<N, +1, 8>(a, stream);
        if constexpr (N <= 512) {
            if (SIGN > 0 && fmode == 7) return launch_line_pass_mode<N, +1, 7>(a, stream);
            if (SIGN > 0 && fmode == 9) return launch_line_pass_mode<N, +1, 9>(a, stream);
        }
    }
    if (fmode >= 6 && fmode <= 9) {
        c21hip_set_error("native FFT: evaluated windows are not built for %d-point lines", N);
        return C21CM_VALUE_ERROR;
    }
    return launch_line_pass_mode<N, SIGN, 0>(a, stream);
}

template <int SIGN>
int dispatch_line_pass(int n, const LinePassArgs &a, int fmode, hipStream_t stream) {
    switch (n) {
        case 64: return launch_line_pass<64, SIGN>(a, fmode, stream);
        case 128: return launch_line_pass<128, SIGN>(a, fmode, stream);
        case 192: return launch_line_pass<192, SIGN>(a, fmode, stream);
        case 256: return launch_line_pass<256, SIGN>(a, fmode, stream);
        case 384: return launch_line_pass<384, SIGN>(a, fmode, stream);
        case 512: return launch_line_pass<512, SIGN>(a, fmode, stream);
        case 768: return launch_line_pass<768, SIGN>(a, fmode, stream);
        case 1024: return launch_line_pass<1024, SIGN>(a, fmode, stream);
        case 1536: return launch_line_pass<1536, SIGN>(a, fmode, stream);
        default:
            c21hip_set_error("native FFT: unsupported line length %d", n);
            return C21CM_VALUE_ERROR;
    }
}

template <int NZ, int EPI = 0>
int launch_z_c2r(const ZPassArgs &a, long nlines, hipStream_t stream) {
    constexpr int H = NZ / 2;
    const float2 *twH = twiddles(H);
    const float2 *twN = twiddles(NZ);
    if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
    const size_t lds = sizeof(float2) * ((size_t)H * (LZ_PLAIN + 1) + H + H / 2 + 1);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)z_c2r_kernel<NZ, EPI>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((z_c2r_kernel<NZ, EPI>), dim3((unsigned)(nlines / LZ_PLAIN)), dim3(kBlock),
                       lds, stream, a, twH, twN);
    LAUNCH_CHECK();
    return 0;
}

// ---- fused pass Z, 512-point lines: wave-level transform -------------------------------------
// H = 256 = 16 x 16.  Sixteen lanes own one line (four lines per wave, sixteen per workgroup);
// lane b holds the points 16a + b, a = 0..15, in registers.  Per grid: Hermitian pre-processing
// (the partner X[H-k] comes back from the line's LDS copy), a 16-point DFT in registers, the
// twiddle w^(bc), one transposition through LDS (rows padded to 17 so that both the column
// write and the row read are conflict free), a second 16-point DFT.  Lane c then holds
// z[c + 16 d], d = 0..15, i.e. cells (2j, 2j+1) for sixteen contiguous j per d across the
// sixteen lanes: the mask is read and written straight from registers.  Everything is
// wave-synchronous -- LDS operations of a wave execute in order -- so there is no workgroup
// barrier in the transform, and a line costs ~80 LDS operations per lane x 16 lanes against
// ~2500 lane-operations in the tile version.
constexpr int ZW_LINES = 16;  // lines per workgroup with 16 lanes per line (P = 16)
// P lanes own a line (P = 16: 512- and 1024-point lines; P = 8 with 16 points per lane: 256-point
// lines); a 256-thread workgroup then holds 256 / P lines.
constexpr int zw_lines(int P) { return kBlock / P; }

// Fused pass Z + f_coll sum + barrier, wave-level transform (A = 16 or 32, see wave_c2r).
// TS: a third grid, the filtered x_e of the spin-temperature run, enters the barrier as
// f_coll zeta > 1 - x_e (IonisationBox.c:1118, clip of :1091-1094).
template <int A, bool TS, int P = 16, bool RC = false>
__global__ void
#if C21X_ZW_OCC
__launch_bounds__(kBlock, (A == 16 && !TS) ? C21X_ZW_OCC : 1)
#else
__launch_bounds__((A == 16 && P == 16) ? C21X_ZW_BLOCK : kBlock, (TS && RC && A == 16 && P == 16) ? C21X_ZW_TSRC_OCC : 1)
#endif
zw_ionise_kernel(ZFusedArgs a, const float2 *__restrict__ twH_global,
                 const float2 *__restrict__ twN_global) {
    constexpr int ZBLK = (A == 16 && P == 16) ? C21X_ZW_BLOCK : kBlock;  // threads of this workgroup
    constexpr int H = P * A, NZ = 2 * H, ZWL = ZBLK / P;
    constexpr int LINE_LDS = A * (P + 1) + 4;  // float2 per line region (padded rows + skew)
    static_assert(LINE_LDS % 2 == 0, "16-byte aligned line regions (mask rows)");
    __shared__ __attribute__((aligned(16))) float2 lines[ZWL * LINE_LDS];
    __shared__ float2 twH[H], twN[H];
    __shared__ double red[ZBLK / 64];
    for (int t = threadIdx.x; t < H; t += ZBLK) {
        twH[t] = twH_global[t];
        twN[t] = twN_global[t];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / P, b = lane % P;
    const int lw = wave * (64 / P) + g;
    // reverse: start where the pass Y that just ran stopped writing, i.e. with the lines the
    // 256 MB Infinity Cache still holds
    const unsigned blk = a.reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const long line = (long)blk * ZWL + lw;
    float2 *L = lines + lw * LINE_LDS;
    const float2 *dm = a.d_main + line * H, *sm = a.s_main + line * H;
    constexpr bool EARLY = (A == 16) && !C21X_ZW_LATE;
    float2 xd[A], xs[A];
#pragma unroll
    for (int q = 0; q < A; q++) xd[q] = dm[P * q + b];
    if (EARLY) {  // both grids in flight from the start; A = 32 has no registers to spare
#pragma unroll
        for (int q = 0; q < A; q++) xs[q] = sm[P * q + b];
    }
    const long lline = logical_line(line, a.ny, a.lb);
    const float dh = a.d_nyq[lline].x, sh = a.s_nyq[lline].x;
    unsigned char *mrow = a.first_cross + lline * NZ;
    // MASK16: the line's mask row as 2 A contiguous bytes per lane (16-byte loads in flight with the
    // spectra); after the transforms the row goes through the line's LDS region, where the lanes pick
    // their (cell 2j, 2j + 1) pairs and leave the changes; changed 16-byte pieces go back to the grid.
    // Sixteen 2-byte loads and up to sixteen 2-byte stores per lane otherwise.
    constexpr bool MASK16 = EARLY && (C21X_ZW_MASK16 || (TS && RC && C21X_ZW_TSRC_LEAN));
    constexpr int MV = MASK16 ? A / 8 : 1;
    uint4 mreg[MV];
    uchar2 old[(EARLY && !MASK16) ? A : 1];
    if constexpr (MASK16) {
#pragma unroll
        for (int v = 0; v < MV; v++) mreg[v] = reinterpret_cast<const uint4 *>(mrow)[b * MV + v];
    } else if (EARLY) {  // mask rows early too
#pragma unroll
        for (int q = 0; q < A; q++)
            old[q] = reinterpret_cast<const uchar2 *>(mrow)[(b + P * (q / P)) + A * (q % P)];
    }
    // RC: the cells' previous N_rec.  Loads inside the barrier loop would each wait for HBM (636
    // instead of ~370 us per radius at 512^3), all sixteen requested here cost the second wave per
    // SIMD (256 VGPRs): the first half is requested now, the second half at the head of the barrier
    // loop, eight iterations ahead of its use.
    // Three lines + N_rec (x_e grid of a spin-temperature run + CELL_RECOMB): 318 registers and one wave
    // per SIMD that way.  STASH: the line's N_rec row is requested as 16-byte pieces with the spectra,
    // parked in LDS after the first transform (2 KB per line) and read from there in the barrier loop.
    constexpr bool STASH = TS && RC && A == 16 && P == 16 && C21X_ZW_TSRC_LEAN;
    constexpr int NRH = (RC && A == 16 && !STASH) ? A / 2 : 1;
    float2 nr_lo[NRH], nr_hi[NRH];
    __shared__ __attribute__((aligned(16))) float nstash[STASH ? ZWL * NZ : 4];
    uint4 nreg[STASH ? NZ / (4 * P) : 1];
    if constexpr (STASH) {
#pragma unroll
        for (int v = 0; v < NZ / (4 * P); v++) nreg[v] = make_uint4(0u, 0u, 0u, 0u);
        if (a.nrec) {
#pragma unroll
            for (int v = 0; v < NZ / (4 * P); v++)
                nreg[v] = reinterpret_cast<const uint4 *>(a.nrec + lline * NZ)[v * P + b];
        }
    } else if constexpr (RC && A == 16) {
        if (a.nrec) {
#pragma unroll
            for (int q = 0; q < NRH; q++)
                nr_lo[q] = reinterpret_cast<const float2 *>(a.nrec + lline * NZ)[(b + P * (q / P)) + A * (q % P)];
        }
    }
    __syncthreads();  // twiddle tables
    if constexpr (STASH) {
        if (a.n_main) {  // fourth line: N_rec filtered at this radius, clipped at zero (IonisationBox.c:808-809)
            float2 xn[A];
            const float2 *nm = a.n_main + line * H;
#pragma unroll
            for (int q = 0; q < A; q++) xn[q] = nm[P * q + b];
            wave_c2r<A, P>(xn, a.n_nyq[lline].x, L, twH, twN, b);
#pragma unroll
            for (int q = 0; q < A; q++) {
                const int j = (b + P * (q / P)) + A * (q % P);
                reinterpret_cast<float2 *>(nstash + lw * NZ)[j] =
                    make_float2(fmaxf(xn[q].x, 0.f), fmaxf(xn[q].y, 0.f));
            }
            wave_fence();
        }
    }
    wave_c2r<A, P>(xd, dh, L, twH, twN, b);
    if constexpr (STASH) {
        if (a.nrec) {
#pragma unroll
            for (int v = 0; v < NZ / (4 * P); v++)
                reinterpret_cast<uint4 *>(nstash + lw * NZ)[v * P + b] = nreg[v];
        }
    }
    if (!EARLY) {
#pragma unroll
        for (int q = 0; q < A; q++) xs[q] = sm[P * q + b];
    }
    wave_fence();
    wave_c2r<A, P>(xs, sh, L, twH, twN, b);
    static_assert(!(TS && RC) || A == 16, "x_e + recombinations: 16 points per lane only (registers)");
    // (RC: the filtered whalo_sfr is only needed where a cell crosses for the first time; it gets
    //  its own pass Z right after this kernel, which looks the crossings of this radius up in the mask)
    float2 xx[TS ? A : 1];
    if constexpr (TS) {  // third grid: x_e
        const float2 *xm = a.x_main + line * H;
#pragma unroll
        for (int q = 0; q < A; q++) xx[q] = xm[P * q + b];
        const float xh = a.x_nyq[lline].x;
        wave_fence();
        wave_c2r<A, P>(xx, xh, L, twH, twN, b);
    }

    if constexpr (RC && A == 16 && !STASH) {
        if (a.nrec) {
#pragma unroll
            for (int q = 0; q < NRH; q++) {
                const int qq = q + NRH;
                nr_hi[q] = reinterpret_cast<const float2 *>(a.nrec + lline * NZ)[(b + P * (qq / P)) + A * (qq % P)];
            }
        }
    }
    if constexpr (MASK16) {
        wave_fence();  // the last transform's reads of the region
#pragma unroll
        for (int v = 0; v < MV; v++) reinterpret_cast<uint4 *>(L)[b * MV + v] = mreg[v];
        wave_fence();
    }
    const double floor_lhs = a.f_limit * a.ion_eff;  // the floored f_coll zeta
    const bool floor_ionises = !TS && a.mass_dep_zeta && (floor_lhs > 1.);
    const float dmin = (float)(-1. + 1e-7);  // IonisationBox.c:803
    double acc = 0.;
#pragma unroll
    for (int q = 0; q < A; q++) {
        // x[P r + d] holds the cells (2j, 2j + 1), j = (b + P r) + A d
        const int j = (b + P * (q / P)) + A * (q % P);
        const float s0 = fmaxf(xs[q].x, 0.f), s1 = fmaxf(xs[q].y, 0.f);
        acc += (double)s0;
        acc += (double)s1;
        double D0 = a.rhocrit_omb * (1. + (double)fmaxf(xd[q].x, dmin));
        double D1 = a.rhocrit_omb * (1. + (double)fmaxf(xd[q].y, dmin));
        bool f0 = false, f1 = false;
        if constexpr (TS) {  // f zeta > 1 - x_e with f = max(s / D, f_limit): both sides times D > 0
            const double n0 = 1. - (double)fminf(fmaxf(xx[q].x, 0.f), 0.999f);
            const double n1 = 1. - (double)fminf(fmaxf(xx[q].y, 0.f), 0.999f);
            f0 = a.mass_dep_zeta && floor_lhs > n0;
            f1 = a.mass_dep_zeta && floor_lhs > n1;
            D0 *= n0;
            D1 *= n1;
        }
        double opd0 = 1., opd1 = 1.;  // RC: 1 + delta_R of the two cells
        if constexpr (RC) {
            // f zeta > (1 - x_e)(1 + rec), rec = N_rec / (1 + delta): both sides times rho (1 + delta) > 0,
            // f = max(s / (rho (1 + delta)), f_limit); without an x_e grid (1 - x_e) = 1
            opd0 = 1. + (double)fmaxf(xd[q].x, dmin);
            opd1 = 1. + (double)fmaxf(xd[q].y, dmin);
            double r0 = a.rec0, r1 = a.rec0;
            if (a.nrec || (STASH && a.n_main)) {
                const float2 nr = STASH ? reinterpret_cast<const float2 *>(nstash + lw * NZ)[j]
                                  : (A == 16) ? (q < NRH ? nr_lo[(A == 16 && !STASH && q < NRH) ? q : 0]
                                                         : nr_hi[(A == 16 && !STASH && q >= NRH) ? q - NRH : 0])
                                              : reinterpret_cast<const float2 *>(a.nrec + lline * NZ)[j];
                r0 = (double)nr.x, r1 = (double)nr.y;
            }
            double n0 = 1., n1 = 1.;
            if constexpr (TS) {
                if (a.x_is_nrec) {  // the third line is N_rec(R), clipped at zero (:808-809)
                    r0 = (double)fmaxf(xx[q].x, 0.f);
                    r1 = (double)fmaxf(xx[q].y, 0.f);
                } else {
                    n0 = 1. - (double)fminf(fmaxf(xx[q].x, 0.f), 0.999f);
                    n1 = 1. - (double)fminf(fmaxf(xx[q].y, 0.f), 0.999f);
                }
            }
            D0 = a.rhocrit_omb * ((opd0 + r0) * n0);
            D1 = a.rhocrit_omb * ((opd1 + r1) * n1);
            f0 = a.mass_dep_zeta && floor_lhs * opd0 > (opd0 + r0) * n0;
            f1 = a.mass_dep_zeta && floor_lhs * opd1 > (opd1 + r1) * n1;
        }
        const bool i0 = (!RC && floor_ionises) || f0 || ((double)s0 * a.ion_eff > D0);
        const bool i1 = (!RC && floor_ionises) || f1 || ((double)s1 * a.ion_eff > D1);
        uchar2 m = MASK16 ? reinterpret_cast<const uchar2 *>(L)[j]
                          : (EARLY ? old[(EARLY && !MASK16) ? q : 0] : reinterpret_cast<const uchar2 *>(mrow)[j]);
        const bool n0 = i0 && m.x == 0, n1 = i1 && m.y == 0;
        if (n0) m.x = (unsigned char)a.r_index;
        if (n1) m.y = (unsigned char)a.r_index;
        if constexpr (MASK16) {
            if (n0 || n1) reinterpret_cast<uchar2 *>(L)[j] = m;
        } else if (n0 || n1 || a.store_all)
            reinterpret_cast<uchar2 *>(mrow)[j] = m;
        if constexpr (RC) {
            // a first crossing leaves its delta_R in the Gamma_12 grid; the whalo_sfr pass of this
            // radius (zw_c2r_kernel, EPI 4) turns it into Gamma_12 = R pref / (1 + delta_R) sfr_R
            float *grow = a.g12 + lline * NZ + 2 * j;
            if (n0 && a.rc != 2) grow[0] = fmaxf(xd[q].x, dmin);
            if (n1 && a.rc != 2) grow[1] = fmaxf(xd[q].y, dmin);
        }
#if C21X_ZW_TSRC_FENCE
        // three lines + N_rec: keep the iterations apart, or the scheduler's hoisting across the
        // unrolled loop costs the second wave per SIMD (318 registers)
        if constexpr (TS && RC) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if constexpr (MASK16) {
        wave_fence();
#pragma unroll
        for (int v = 0; v < MV; v++) {
            const uint4 n = reinterpret_cast<const uint4 *>(L)[b * MV + v], o = mreg[v];
            if (a.store_all || n.x != o.x || n.y != o.y || n.z != o.z || n.w != o.w)
                reinterpret_cast<uint4 *>(mrow)[b * MV + v] = n;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.;
#pragma unroll
        for (int w = 0; w < ZBLK / 64; w++) sum += red[w];
        a.partials[blk] = sum;
    }
}

// Plain pass Z (one grid) on the wave-level transform, with the store epilogues of z_c2r_kernel
// (EPI 0 store / divide, 1 store + extrema, 2 closed-form f_coll + sum, 3 floor-scale store +
// statistics).  Sixteen lines per workgroup like the tile kernel, so the partial arrays have the
// same length either way.
template <int A, int EPI, int P = 16>
__global__ void __launch_bounds__(kBlock)
zw_c2r_kernel(ZPassArgs a, const float2 *__restrict__ twH_global,
              const float2 *__restrict__ twN_global) {
    constexpr int H = P * A, NZ = 2 * H, ZWL = zw_lines(P);
    constexpr int LINE_LDS = A * (P + 1) + 4;
    static_assert(LINE_LDS % 2 == 0, "16-byte aligned line regions (EPI 7)");
    __shared__ __attribute__((aligned(16))) float2 lines[ZWL * LINE_LDS];
    __shared__ float2 twH[H], twN[H];
    for (int t = threadIdx.x; t < H; t += kBlock) {
        twH[t] = twH_global[t];
        twN[t] = twN_global[t];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / P, b = lane % P;
    const int lw = wave * (64 / P) + g;
    // reverse: start with the lines the pass Y that just ran wrote last (Infinity Cache), and leave the
    // head of the real grid there for the sweep that follows (Eulerian loops)
    const unsigned blk = a.reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const long line = (long)blk * ZWL + lw;
    float2 *L = lines + lw * LINE_LDS;
    const float2 *src = a.main + line * H;
    float2 x[A];
#pragma unroll
    for (int q = 0; q < A; q++) x[q] = src[P * q + b];
    const long lline = logical_line(line, a.ny, a.lb);
    const float xh = a.nyq[lline].x;
    float2 fprev[EPI == 6 ? A : 1];
    uchar2 mprev[EPI == 6 ? A : 1];
    // EPI 7: the mask row of the line as 2 A contiguous bytes per lane (16-byte loads, in flight under
    // the transform); the lanes' (cell 2j, 2j + 1) pairs are picked out of the line's LDS region afterwards
    // (EPI 4 reads the rows the same way -- sixteen 2-byte loads per lane issued after the transform
    //  made the Gamma_12 pass slower than a pass Z that stores its whole grid: 0.30 against 0.25 ms)
    constexpr bool MROW = (EPI == 7 || EPI == 8 || EPI == 9 || (EPI == 4 && C21X_EPI4_MASK16));
    constexpr int MV = MROW ? A / 8 : 1;
    uint4 mreg[MV];
    if constexpr (MROW) {
        const uint4 *mrow = reinterpret_cast<const uint4 *>(((EPI == 4) ? a.mask : a.mask_rw) + lline * NZ) + b * MV;
#pragma unroll
        for (int v = 0; v < MV; v++) mreg[v] = mrow[v];
    }
    __shared__ float ftab[(EPI == 8 || EPI == 9) ? C21CM_NDELTA_TABLE : 1];
    float2 dreg[EPI == 8 ? A : 1];
    if constexpr (EPI == 9)  // the radius' f_coll table (the filtered density is this transform's own output)
        for (int t = threadIdx.x; t < C21CM_NDELTA_TABLE; t += kBlock) ftab[t] = a.table[t];
    if constexpr (EPI == 8) {  // the radius' f_coll table and the cells' filtered density
        for (int t = threadIdx.x; t < C21CM_NDELTA_TABLE; t += kBlock) ftab[t] = a.table[t];
#pragma unroll
        for (int q = 0; q < A; q++) {
            const int j = (b + P * (q / P)) + A * (q % P);
            dreg[q] = reinterpret_cast<const float2 *>(a.delta_fil + lline * a.out_zstride)[j];
        }
    }
    if constexpr (EPI == 6) {  // the previous radius' f_coll and the mask rows, in flight under the transform
#pragma unroll
        for (int q = 0; q < A; q++) {
            const int j = (b + P * (q / P)) + A * (q % P);
            fprev[q] = reinterpret_cast<const float2 *>(a.f_prev + lline * NZ)[j];
            mprev[q] = reinterpret_cast<const uchar2 *>(a.mask_rw + lline * NZ)[j];
        }
    }
    __syncthreads();  // twiddle tables
    wave_c2r<A, P>(x, xh, L, twH, twN, b);
    if constexpr (MROW) {
        wave_fence();  // the transform's last reads of the region
#pragma unroll
        for (int v = 0; v < MV; v++) reinterpret_cast<uint4 *>(L)[b * MV + v] = mreg[v];
        wave_fence();
    }

    double acc0 = 0., acc1 = 0., acc2 = 0.;
    float t_sure = 0.f, t_maybe = 0.f;
    if constexpr (EPI == 7 || EPI == 9) {
        t_sure = (float)a.band[0];
        t_maybe = (float)a.band[1];
    }
#pragma unroll
    for (int q = 0; q < A; q++) {
        const int j = (b + P * (q / P)) + A * (q % P);  // cells (2j, 2j + 1)
        float2 v = x[q];
        if (a.out_scale != 1.0f) {
            v.x *= a.out_scale;
            v.y *= a.out_scale;
        }
        if (EPI == 0 && a.out_div != 0.f) {
            v.x = __fdiv_rn(v.x, a.out_div);
            v.y = __fdiv_rn(v.y, a.out_div);
        }
        if (EPI == 0 && a.out_floor) {
            if ((double)v.x < -1.0 + 1e-7) v.x = (float)(-1.0 + 1e-7);
            if ((double)v.y < -1.0 + 1e-7) v.y = (float)(-1.0 + 1e-7);
        }
        if (EPI == 3) {
            if ((double)v.x < a.min_value) v.x = (float)a.min_value;
            if ((double)v.y < a.min_value) v.y = (float)a.min_value;
            v.x = (float)((double)v.x * a.const_factor);
            v.y = (float)((double)v.y * a.const_factor);
            acc2 += (double)v.x;
            acc2 += (double)v.y;
        }
        if (EPI == 5) {
            const double mean_fix = a.fix_mean ? a.mean_f_coll / *a.mean_dev : 1.;
            const float2 f = reinterpret_cast<const float2 *>(a.f_out + lline * NZ)[j];
            uchar2 m = reinterpret_cast<const uchar2 *>(a.mask_rw + lline * NZ)[j];
            double c0 = mean_fix * (double)f.x, c1 = mean_fix * (double)f.y;
            if (a.mass_dep_zeta && c0 < a.f_limit) c0 = a.f_limit;
            if (a.mass_dep_zeta && c1 < a.f_limit) c1 = a.f_limit;
            const double x0 = (double)fminf(fmaxf(v.x, 0.f), 0.999f), x1 = (double)fminf(fmaxf(v.y, 0.f), 0.999f);
            const bool h0 = c0 * a.ion_eff > (1. - x0) && m.x == 0, h1 = c1 * a.ion_eff > (1. - x1) && m.y == 0;
            if (h0) m.x = (unsigned char)a.r_index;
            if (h1) m.y = (unsigned char)a.r_index;
            if (h0 || h1) reinterpret_cast<uchar2 *>(a.mask_rw + lline * NZ)[j] = m;
        } else if (EPI == 4) {
            // (requesting the crossings' delta_R before the transform was tried: 0.289 against 0.252 ms)
            const uchar2 m = MROW ? reinterpret_cast<const uchar2 *>(L)[j]
                                  : reinterpret_cast<const uchar2 *>(a.mask + lline * NZ)[j];
            float *grow = a.out + lline * NZ + 2 * j;
            if (m.x == (unsigned char)a.r_index)
                grow[0] = (float)(a.const_factor / (1. + (double)grow[0]) * (double)fmaxf(v.x, 0.f));
            if (m.y == (unsigned char)a.r_index)
                grow[1] = (float)(a.const_factor / (1. + (double)grow[1]) * (double)fmaxf(v.y, 0.f));
        } else if (EPI == 8) {
            const float2 dd = dreg[EPI == 8 ? q : 0];
            const float d0 = clip_delta_eulerian(dd.x), d1 = clip_delta_eulerian(dd.y);
            const double inv_w = 1. / a.tab_width;  // (loop invariant: hoisted)
            double f0 = eval_table_f_inv((double)d0, a.tab_min, a.tab_width, inv_w, ftab);
            double f1 = eval_table_f_inv((double)d1, a.tab_min, a.tab_width, inv_w, ftab);
            if (a.tab_mode != C21CM_FCOLL_TABLE_LINEAR) {
                f0 = exp_f32acc(f0);
                f1 = exp_f32acc(f1);
            }
            acc0 += f0;
            acc0 += f1;
            const float g0 = (float)f0, g1 = (float)f1;  // what the dense grid would hold
            const float xc0 = fminf(fmaxf(v.x, 0.f), 0.999f), xc1 = fminf(fmaxf(v.y, 0.f), 0.999f);
            uchar2 m = reinterpret_cast<const uchar2 *>(L)[j];
            float *frow = a.f_out + lline * NZ + 2 * j, *xrow = a.xe_pend + lline * NZ + 2 * j;
            bool ch = false;
            if (a.r_prev >= 0 && (m.x == 255 || m.y == 255)) {  // the previous radius' undecided cells
                const double mf_prev = *a.mean_dev;
                if (m.x == 255) {
                    double c0 = mf_prev * (double)frow[0];
                    if (a.mass_dep_zeta && c0 < a.f_limit) c0 = a.f_limit;
                    m.x = (c0 * a.ion_eff > (1. - (double)xrow[0])) ? (unsigned char)a.r_prev : (unsigned char)0;
                }
                if (m.y == 255) {
                    double c1 = mf_prev * (double)frow[1];
                    if (a.mass_dep_zeta && c1 < a.f_limit) c1 = a.f_limit;
                    m.y = (c1 * a.ion_eff > (1. - (double)xrow[1])) ? (unsigned char)a.r_prev : (unsigned char)0;
                }
                ch = true;
            }
            if (m.x == 0 || m.y == 0) {
                const double mf_lo = a.band[0], mf_hi = a.band[1];
                double l0 = mf_lo * (double)g0, l1 = mf_lo * (double)g1;
                double h0 = mf_hi * (double)g0, h1 = mf_hi * (double)g1;
                if (a.mass_dep_zeta) {
                    if (l0 < a.f_limit) l0 = a.f_limit;
                    if (l1 < a.f_limit) l1 = a.f_limit;
                    if (h0 < a.f_limit) h0 = a.f_limit;
                    if (h1 < a.f_limit) h1 = a.f_limit;
                }
                const double n0 = 1. - (double)xc0, n1 = 1. - (double)xc1;
                if (m.x == 0 && h0 * a.ion_eff > n0) {
                    const bool sure = l0 * a.ion_eff > n0;
                    m.x = sure ? (unsigned char)a.r_index : (unsigned char)255;
                    if (!sure) frow[0] = g0, xrow[0] = xc0;
                    ch = true;
                }
                if (m.y == 0 && h1 * a.ion_eff > n1) {
                    const bool sure = l1 * a.ion_eff > n1;
                    m.y = sure ? (unsigned char)a.r_index : (unsigned char)255;
                    if (!sure) frow[1] = g1, xrow[1] = xc1;
                    ch = true;
                }
            }
            if (ch) reinterpret_cast<uchar2 *>(L)[j] = m;
        } else if (EPI == 7 || EPI == 9) {
            double f0, f1;
            if constexpr (EPI == 9) {  // fcoll_eulerian_band_kernel's statements on the value the store pass would have written
                const double inv_w = 1. / a.tab_width;
                f0 = eval_table_f_inv((double)clip_delta_eulerian(v.x), a.tab_min, a.tab_width, inv_w, ftab);
                f1 = eval_table_f_inv((double)clip_delta_eulerian(v.y), a.tab_min, a.tab_width, inv_w, ftab);
                if (a.tab_mode != C21CM_FCOLL_TABLE_LINEAR) {
                    f0 = exp_f32acc(f0);
                    f1 = exp_f32acc(f1);
                }
            } else {
                f0 = (a.sig < 0) ? 0. : fgtrm_bias_fast_inv(clip_delta_eulerian(v.x), a.sig, a.delta_c);
                f1 = (a.sig < 0) ? 0. : fgtrm_bias_fast_inv(clip_delta_eulerian(v.y), a.sig, a.delta_c);
            }
            acc0 += f0;
            acc0 += f1;
            const float g0 = (float)f0, g1 = (float)f1;  // what the dense grid of EPI 2 would hold
            uchar2 m = reinterpret_cast<const uchar2 *>(L)[j];
            float *frow = a.f_out + lline * NZ + 2 * j;
            bool ch = false;
            if (a.r_prev >= 0 && (m.x == 255 || m.y == 255)) {  // the previous radius' undecided cells
                const float t_prev = (float)*a.mean_dev;
                if (m.x == 255) m.x = (frow[0] >= t_prev) ? (unsigned char)a.r_prev : (unsigned char)0;
                if (m.y == 255) m.y = (frow[1] >= t_prev) ? (unsigned char)a.r_prev : (unsigned char)0;
                ch = true;
            }
            if (m.x == 0 && g0 >= t_maybe) {
                const bool sure = g0 >= t_sure;
                m.x = sure ? (unsigned char)a.r_index : (unsigned char)255;
                if (!sure) frow[0] = g0;
                ch = true;
            }
            if (m.y == 0 && g1 >= t_maybe) {
                const bool sure = g1 >= t_sure;
                m.y = sure ? (unsigned char)a.r_index : (unsigned char)255;
                if (!sure) frow[1] = g1;
                ch = true;
            }
            if (ch) reinterpret_cast<uchar2 *>(L)[j] = m;
        } else if (EPI == 2 || EPI == 6) {
            const double f0 = (a.sig < 0) ? 0. : fgtrm_bias_fast_inv(clip_delta_eulerian(v.x), a.sig, a.delta_c);
            const double f1 = (a.sig < 0) ? 0. : fgtrm_bias_fast_inv(clip_delta_eulerian(v.y), a.sig, a.delta_c);
            acc0 += f0;
            acc0 += f1;
            reinterpret_cast<float2 *>(a.f_out + lline * NZ)[j] = make_float2((float)f0, (float)f1);
            if constexpr (EPI == 6) {  // eulerian_mask_kernel's statements for the previous radius
                const double mean_fix = a.fix_mean ? a.mean_f_coll / *a.mean_dev : 1.;
                const float2 f = fprev[EPI == 6 ? q : 0];
                uchar2 m = mprev[EPI == 6 ? q : 0];
                double c0 = mean_fix * (double)f.x, c1 = mean_fix * (double)f.y;
                if (a.mass_dep_zeta && c0 < a.f_limit) c0 = a.f_limit;
                if (a.mass_dep_zeta && c1 < a.f_limit) c1 = a.f_limit;
                const bool h0 = c0 * a.ion_eff > (1. - 0.) && m.x == 0, h1 = c1 * a.ion_eff > (1. - 0.) && m.y == 0;
                if (h0) m.x = (unsigned char)a.r_index;
                if (h1) m.y = (unsigned char)a.r_index;
                if (h0 || h1) reinterpret_cast<uchar2 *>(a.mask_rw + lline * NZ)[j] = m;
            }
        } else {
            if (EPI != 10) reinterpret_cast<float2 *>(a.out + lline * a.out_zstride)[j] = v;  // (10: the extrema alone)
            if (EPI == 1 || EPI == 3 || EPI == 10) {
                const double lo = fmin((double)v.x, (double)v.y), hi = fmax((double)v.x, (double)v.y);
                acc0 = (q == 0) ? lo : fmin(acc0, lo);
                acc1 = (q == 0) ? hi : fmax(acc1, hi);
            }
        }
    }
    if constexpr (EPI == 7 || EPI == 8 || EPI == 9) {  // changed 16-byte pieces of the mask row back to the grid
        wave_fence();
#pragma unroll
        for (int v = 0; v < MV; v++) {
            const uint4 n = reinterpret_cast<const uint4 *>(L)[b * MV + v], o = mreg[v];
            if (n.x != o.x || n.y != o.y || n.z != o.z || n.w != o.w)
                reinterpret_cast<uint4 *>(a.mask_rw + lline * NZ)[b * MV + v] = n;
        }
    }
    if (EPI != 0 && EPI != 4 && EPI != 5) {
        __shared__ double red0[kBlock / 64], red1[kBlock / 64], red2[kBlock / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o0 = __shfl_down(acc0, off, 64), o1 = __shfl_down(acc1, off, 64);
            acc0 = (EPI == 2 || EPI == 6 || EPI == 7 || EPI == 8 || EPI == 9) ? acc0 + o0 : fmin(acc0, o0);
            acc1 = fmax(acc1, o1);
            if (EPI == 3) acc2 += __shfl_down(acc2, off, 64);
        }
        if (lane == 0) {
            red0[wave] = acc0;
            red1[wave] = acc1;
            red2[wave] = acc2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double r0 = red0[0], r1 = red1[0], r2 = red2[0];
#pragma unroll
            for (int w = 1; w < kBlock / 64; w++) {
                r0 = (EPI == 2 || EPI == 6 || EPI == 7 || EPI == 8 || EPI == 9) ? r0 + red0[w] : fmin(r0, red0[w]);
                r1 = fmax(r1, red1[w]);
                r2 += red2[w];
            }
            a.p0[blk] = r0;
            if (EPI == 1 || EPI == 3 || EPI == 10) a.p1[blk] = r1;
            if (EPI == 3) a.p2[blk] = r2;
        }
    }
}

// ---- forward pass Z (real -> complex) on the wave-level transform ----------------------------
// The mirror image of wave_c2r.  In: x[16 r + d] = z[(b + 16 r) + A d], z[j] = (cell 2j, cell
// 2j+1) -- the layout wave_c2r ends in, so the loads are 128-byte runs.  A 16-point DFT over d in
// registers, the twiddle exp(-2 pi i c b'/H), one transposition through the 17-padded LDS
// region, a DFT over the A values of c, then the Hermitian post-processing whose partner
// Zf[H-k] comes back from the line's LDS copy.  Out: x[a] = X[16 a + b] (k < H), *xh = X[H].
template <int A, int P = 16>
__device__ __forceinline__ void wave_r2c(float2 (&x)[A], float *xh, float2 *L, const float2 *twH,
                                         const float2 *twN, int b) {
    static_assert(A % P == 0, "rows per lane");
    constexpr int H = P * A;
#pragma unroll
    for (int r = 0; r < A / P; r++) {
        Dft<P, -1>::run(x + P * r);  // over d: Y_c[b'], c = b + P r
        const int c = b + P * r;
#pragma unroll
        for (int bb = 0; bb < P; bb++)
            L[c * (P + 1) + bb] = (bb == 0) ? x[P * r] : cmul(x[P * r + bb], twH[c * bb]);
    }
    wave_fence();
#pragma unroll
    for (int c = 0; c < A; c++) x[c] = L[c * (P + 1) + b];
    Dft<A, -1>::run(x);  // over c: Zf[P a + b]
    wave_fence();        // the column reads are done before the region is overwritten
#pragma unroll
    for (int a = 0; a < A; a++) L[a * (P + 1) + b] = x[a];
    wave_fence();
#pragma unroll
    for (int a = 0; a < A; a++) {
        const int k = P * a + b;
        const int kp = (H - k) & (H - 1);
        const float2 Z = x[a], B = L[(kp / P) * (P + 1) + (kp % P)];
        if (k == 0) {
            x[a] = make_float2(Z.x + Z.y, 0.f);
            *xh = Z.x - Z.y;
        } else {
            const float2 E = make_float2(0.5f * (Z.x + B.x), 0.5f * (Z.y - B.y));
            const float2 D = make_float2(0.5f * (Z.x - B.x), 0.5f * (Z.y + B.y));
            const float2 wD = cmul(D, twN[k]);
            x[a] = make_float2(E.x + wD.y, E.y - wD.x);  // E - i w_k D
        }
    }
}

template <int A, int P = 16>
__global__ void __launch_bounds__(kBlock)
zw_r2c_kernel(ZFwdArgs a, const float2 *__restrict__ twH_global,
              const float2 *__restrict__ twN_global) {
    constexpr int H = P * A, ZWL = zw_lines(P);
    constexpr int LINE_LDS = A * (P + 1) + 4;
    __shared__ float2 lines[ZWL * LINE_LDS];
    __shared__ float2 twH[H], twN[H];
    for (int t = threadIdx.x; t < H; t += kBlock) {
        twH[t] = twH_global[t];
        twN[t] = twN_global[t];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / P, b = lane % P;
    const int lw = wave * (64 / P) + g;
    const long line = (long)blockIdx.x * ZWL + lw;
    const long lline = logical_line(line, a.ny, a.lb);
    float2 *L = lines + lw * LINE_LDS;
    const float2 *src = reinterpret_cast<const float2 *>(a.in + lline * a.in_zstride);
    const bool clip = a.lo <= a.hi;
    float2 x[A];
#pragma unroll
    for (int q = 0; q < A; q++) {
        const int j = (b + P * (q / P)) + A * (q % P);
        float2 v = src[j];
        if (clip) {
            v.x = (float)fmax(fmin((double)v.x * a.factor, a.hi), a.lo);
            v.y = (float)fmax(fmin((double)v.y * a.factor, a.hi), a.lo);
        } else if (a.factor != 1.0) {
            v.x = (float)((double)v.x * a.factor);
            v.y = (float)((double)v.y * a.factor);
        }
        x[q] = v;
    }
    __syncthreads();  // twiddle tables
    float xh = 0.f;
    wave_r2c<A, P>(x, &xh, L, twH, twN, b);
    float2 *dst = a.main + line * H;
#pragma unroll
    for (int q = 0; q < A; q++) dst[P * q + b] = x[q];
    if (b == 0) a.nyq[lline] = make_float2(xh, 0.f);
}

// ---- 1024-point z-lines: three radix-8 stages across a whole wave -----------------------------
// H = 512 = 8 x 8 x 8.  One wave owns a line, lane p holds X[64 a + p], a < 8: eight values per
// grid instead of the 32 of the (A = 32, P = 16) scheme, whose radix-32 butterflies need all 256
// VGPRs and run one wave per SIMD.  Three 8-point DFTs in registers, two transpositions through
// the line's LDS region in between (plus the Hermitian pre-processing's partner exchange):
//   k = 64 a + p,  j = j0 + 8 j1 + 64 j2
//   stage 1 (lane p):          Y_p[j0]      = sum_a X[64 a + p] w64^(a j0)        x w^(j0 p)
//   stage 2 (lane 8 j0 + c):   V_j0,c[j1]   = sum_b T_j0[8 b + c] w8^(b j1)       x w^(8 c j1)
//   stage 3 (lane j0 + 8 j1):  z[j0 + 8 j1 + 64 j2] = sum_c V_j0,c[j1] w8^(c j2)
// so lane p ends with z[p + 64 j2], j2 < 8: natural order, 512-byte runs per j2.
// In: x[a] = X[64 a + p], xh = Re X[H].  L: 576 float2 of LDS for this line.
__device__ __forceinline__ void wave3_c2r(float2 (&x)[8], float xh, float2 *L, const float2 *twH,
                                          const float2 *twN, int p) {
    constexpr int H = 512, M1 = 72, M2 = 9;
#pragma unroll
    for (int a = 0; a < 8; a++) L[a * M1 + p] = x[a];
    wave_fence();
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const int k = 64 * a + p;
        const int kp = (H - k) & (H - 1);
        const float2 Xk = x[a];
        float2 B = L[(kp >> 6) * M1 + (kp & 63)];
        if (k == 0) B = make_float2(xh, 0.f);
        const float2 E = make_float2(Xk.x + B.x, Xk.y - B.y);
        const float2 D = make_float2(Xk.x - B.x, Xk.y + B.y);
        float2 w = twN[k];
        w.y = -w.y;
        const float2 O = cmul(D, w);
        x[a] = (k == 0) ? make_float2(Xk.x + xh, Xk.x - xh) : make_float2(E.x - O.y, E.y + O.x);
    }
    Dft<8, +1>::run(x);  // over a: Y_p[j0]
    wave_fence();        // partner reads done
#pragma unroll
    for (int j0 = 0; j0 < 8; j0++) {
        float2 w = twH[j0 * p];
        w.y = -w.y;
        L[j0 * M1 + p] = (j0 == 0) ? x[0] : cmul(x[j0], w);
    }
    wave_fence();
    {
        const int j0 = p >> 3, c = p & 7;
#pragma unroll
        for (int b = 0; b < 8; b++) x[b] = L[j0 * M1 + 8 * b + c];
        Dft<8, +1>::run(x);  // over b: V_j0,c[j1]
        wave_fence();
#pragma unroll
        for (int j1 = 0; j1 < 8; j1++) {
            float2 w = twH[8 * c * j1];
            w.y = -w.y;
            L[p * M2 + j1] = (j1 == 0) ? x[0] : cmul(x[j1], w);
        }
    }
    wave_fence();
    {
        const int j0 = p & 7, j1 = p >> 3;
#pragma unroll
        for (int c = 0; c < 8; c++) x[c] = L[(8 * j0 + c) * M2 + j1];
        Dft<8, +1>::run(x);  // over c: z[p + 64 j2]
    }
}

// The forward counterpart (round 4): the mirror image of wave3_c2r, same LDS arrangements walked
// backwards.  j = j0 + 8 j1 + 64 j2, k = c + 8 b + 64 a, w = exp(-2 pi i / 512):
//   stage 1 (lane j0 + 8 j1):  U_j0,j1[c] = sum_j2 z[j] w8^(j2 c)                 x w^(8 j1 c)
//   stage 2 (lane 8 j0 + c):   V_j0,c[b]  = sum_j1 U_j0,j1[c] w8^(j1 b)           x w^(j0 (c + 8 b))
//   stage 3 (lane c + 8 b):    Zf[64 a + c + 8 b] = sum_j0 V_j0,c[b] w8^(j0 a)
// then the Hermitian post-processing of wave_r2c with the partner Zf[H - k] from the line's LDS copy.
// In: x[j2] = z[p + 64 j2] (512-byte runs).  Out: x[a] = X[64 a + p] (k < H), *xh = X[H] -- the layout
// wave3_c2r starts from.  Eight values per lane instead of the 32 of wave_r2c<32, 16>, whose radix-32
// butterflies fill the register file (one wave per SIMD: 3.55 ms per 1024^3 grid, 2.4 TB/s).
__device__ __forceinline__ void wave3_r2c(float2 (&x)[8], float *xh, float2 *L, const float2 *twH,
                                          const float2 *twN, int p) {
    constexpr int H = 512, M1 = 72, M2 = 9;
    Dft<8, -1>::run(x);  // over j2: U[c]
    {
        const int j0 = p & 7, j1 = p >> 3;
#pragma unroll
        for (int c = 0; c < 8; c++)
            L[(8 * j0 + c) * M2 + j1] = (c == 0) ? x[0] : cmul(x[c], twH[8 * j1 * c]);
    }
    wave_fence();
    {
        const int j0 = p >> 3, c = p & 7;
#pragma unroll
        for (int j1 = 0; j1 < 8; j1++) x[j1] = L[p * M2 + j1];
        Dft<8, -1>::run(x);  // over j1: V[b]
        wave_fence();        // the reads of the M2 arrangement are done before the M1 one overwrites it
#pragma unroll
        for (int b = 0; b < 8; b++) L[j0 * M1 + 8 * b + c] = cmul(x[b], twH[j0 * (c + 8 * b)]);
    }
    wave_fence();
#pragma unroll
    for (int j0 = 0; j0 < 8; j0++) x[j0] = L[j0 * M1 + p];
    Dft<8, -1>::run(x);  // over j0: Zf[64 a + p]
    wave_fence();
#pragma unroll
    for (int a = 0; a < 8; a++) L[a * M1 + p] = x[a];
    wave_fence();
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const int k = 64 * a + p;
        const int kp = (H - k) & (H - 1);
        const float2 Z = x[a], B = L[(kp >> 6) * M1 + (kp & 63)];
        if (k == 0) {
            x[a] = make_float2(Z.x + Z.y, 0.f);
            *xh = Z.x - Z.y;
        } else {
            const float2 E = make_float2(0.5f * (Z.x + B.x), 0.5f * (Z.y - B.y));
            const float2 D = make_float2(0.5f * (Z.x - B.x), 0.5f * (Z.y + B.y));
            const float2 wD = cmul(D, twN[k]);
            x[a] = make_float2(E.x + wD.y, E.y - wD.x);  // E - i w_k D
        }
    }
}

// Forward pass Z of 1024-point z-lines on wave3_r2c: LPW lines per wave one after the other (the loads
// of all of them in flight first), four waves per workgroup.
template <int LPW>
__global__ void __launch_bounds__(kBlock)
zw3_r2c_kernel(ZFwdArgs a, const float2 *__restrict__ twH_global,
               const float2 *__restrict__ twN_global) {
    constexpr int H = 512, LINE_LDS = 576 + 8;
    __shared__ float2 lines[(kBlock / 64) * LINE_LDS];
    __shared__ float2 twH[H], twN[H];
    for (int t = threadIdx.x; t < H; t += kBlock) {
        twH[t] = twH_global[t];
        twN[t] = twN_global[t];
    }
    const int p = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *L = lines + wave * LINE_LDS;
    const bool clip = a.lo <= a.hi;
    float2 x[LPW][8];
    long line[LPW], lline[LPW];
#pragma unroll
    for (int u = 0; u < LPW; u++) {
        line[u] = ((long)blockIdx.x * (kBlock / 64) + wave) * LPW + u;
        lline[u] = logical_line(line[u], a.ny, a.lb);
        const float2 *src = reinterpret_cast<const float2 *>(a.in + lline[u] * a.in_zstride);
#pragma unroll
        for (int q = 0; q < 8; q++) x[u][q] = src[p + 64 * q];
    }
    __syncthreads();  // twiddle tables
#pragma unroll
    for (int u = 0; u < LPW; u++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            float2 v = x[u][q];
            if (clip) {
                v.x = (float)fmax(fmin((double)v.x * a.factor, a.hi), a.lo);
                v.y = (float)fmax(fmin((double)v.y * a.factor, a.hi), a.lo);
            } else if (a.factor != 1.0) {
                v.x = (float)((double)v.x * a.factor);
                v.y = (float)((double)v.y * a.factor);
            }
            x[u][q] = v;
        }
        float xh = 0.f;
        if (u > 0) wave_fence();  // the previous line's partner reads
        wave3_r2c(x[u], &xh, L, twH, twN, p);
        float2 *dst = a.main + line[u] * H;
#pragma unroll
        for (int q = 0; q < 8; q++) dst[64 * q + p] = x[u][q];
        if (p == 0) a.nyq[lline[u]] = make_float2(xh, 0.f);
    }
}

// Fused pass Z + f_coll sum + barrier for 1024-point z-lines on wave3_c2r: four lines per
// workgroup, each lane ends with the cells (2j, 2j+1), j = p + 64 j2.
template <bool TS>
__global__ void __launch_bounds__(kBlock)
zw3_ionise_kernel(ZFusedArgs a, const float2 *__restrict__ twH_global,
                  const float2 *__restrict__ twN_global) {
    constexpr int H = 512, NZ = 1024, LINE_LDS = 576 + 8;
    __shared__ float2 lines[(kBlock / 64) * LINE_LDS];
    __shared__ float2 twH[H], twN[H];
    __shared__ double red[kBlock / 64];
    for (int t = threadIdx.x; t < H; t += kBlock) {
        twH[t] = twH_global[t];
        twN[t] = twN_global[t];
    }
    const int p = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long line = (long)blockIdx.x * (kBlock / 64) + wave;
    float2 *L = lines + wave * LINE_LDS;
    const float2 *dm = a.d_main + line * H, *sm = a.s_main + line * H;
    float2 xd[8], xs[8];
#pragma unroll
    for (int q = 0; q < 8; q++) xd[q] = dm[64 * q + p];
#pragma unroll
    for (int q = 0; q < 8; q++) xs[q] = sm[64 * q + p];
    const long lline = logical_line(line, a.ny, a.lb);
    const float dh = a.d_nyq[lline].x, sh = a.s_nyq[lline].x;
    unsigned char *mrow = a.first_cross + lline * NZ;
    uchar2 old[8];
#pragma unroll
    for (int q = 0; q < 8; q++) old[q] = reinterpret_cast<const uchar2 *>(mrow)[p + 64 * q];
    float2 xx[TS ? 8 : 1];
    float xeh = 0.f;
    if constexpr (TS) {
        const float2 *xm = a.x_main + line * H;
#pragma unroll
        for (int q = 0; q < 8; q++) xx[q] = xm[64 * q + p];
        xeh = a.x_nyq[lline].x;
    }
    __syncthreads();  // twiddle tables
    wave3_c2r(xd, dh, L, twH, twN, p);
    wave_fence();
    wave3_c2r(xs, sh, L, twH, twN, p);
    if constexpr (TS) {
        wave_fence();
        wave3_c2r(xx, xeh, L, twH, twN, p);
    }
    const double floor_lhs = a.f_limit * a.ion_eff;
    const bool floor_ionises = !TS && a.mass_dep_zeta && (floor_lhs > 1.);
    const float dmin = (float)(-1. + 1e-7);  // IonisationBox.c:803
    double acc = 0.;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int j = p + 64 * q;
        const float s0 = fmaxf(xs[q].x, 0.f), s1 = fmaxf(xs[q].y, 0.f);
        acc += (double)s0;
        acc += (double)s1;
        double D0 = a.rhocrit_omb * (1. + (double)fmaxf(xd[q].x, dmin));
        double D1 = a.rhocrit_omb * (1. + (double)fmaxf(xd[q].y, dmin));
        bool f0 = false, f1 = false;
        if constexpr (TS) {
            const double n0 = 1. - (double)fminf(fmaxf(xx[q].x, 0.f), 0.999f);
            const double n1 = 1. - (double)fminf(fmaxf(xx[q].y, 0.f), 0.999f);
            f0 = a.mass_dep_zeta && floor_lhs > n0;
            f1 = a.mass_dep_zeta && floor_lhs > n1;
            D0 *= n0;
            D1 *= n1;
        }
        const bool i0 = floor_ionises || f0 || ((double)s0 * a.ion_eff > D0);
        const bool i1 = floor_ionises || f1 || ((double)s1 * a.ion_eff > D1);
        uchar2 m = old[q];
        const bool n0 = i0 && m.x == 0, n1 = i1 && m.y == 0;
        if (n0) m.x = (unsigned char)a.r_index;
        if (n1) m.y = (unsigned char)a.r_index;
        if (n0 || n1 || a.store_all) reinterpret_cast<uchar2 *>(mrow)[j] = m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (p == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) sum += red[w];
        a.partials[blockIdx.x] = sum;
    }
}

bool zw_enabled() {
    static int cached = -1;
    if (cached < 0) {
        const char *e = getenv("C21CM_ZPASS");
        cached = (e && e[0] == 't') ? 0 : 1;  // C21CM_ZPASS=tile selects the tile version
    }
    return cached == 1;
}

// 1024-point fused pass Z: the three-stage kernel (C21CM_ZPASS=wave32 keeps the A = 32 one)
bool zw3_selected(int nz, long nlines) {
    static int w32 = -1;
    if (w32 < 0) {
        const char *e = getenv("C21CM_ZPASS");
        w32 = (e && e[0] == 'w') ? 1 : 0;
    }
    return nz == 1024 && zw_enabled() && !w32 && nlines % (kBlock / 64) == 0;
}

// lines per workgroup of the wave-level kernels at this z-line length (0: not covered)
int zw_lines_of(int nz, long nlines) {
    if (!zw_enabled()) return 0;
    const int l = (nz == 512 || nz == 1024) ? zw_lines(16) : (nz == 256 ? zw_lines(8) : 0);
    return (l && nlines % l == 0) ? l : 0;
}

template <int NZ>
int launch_z_fused(const ZFusedArgs &a, long nlines, hipStream_t stream) {
    constexpr int H = NZ / 2;
    const float2 *twH = twiddles(H);
    const float2 *twN = twiddles(NZ);
    if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
    const size_t lds = sizeof(float2) * ((size_t)H * (LZ_FUSED + 1) + H + H / 2 + 1);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)z_c2r_ionise_kernel<NZ>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((z_c2r_ionise_kernel<NZ>), dim3((unsigned)(nlines / LZ_FUSED)), dim3(kBlock), lds,
                       stream, a, twH, twN);
    LAUNCH_CHECK();
    return 0;
}

// *n_partials: how many workgroup partials of sum(stars) the launch wrote
int dispatch_z_fused(int nz, const ZFusedArgs &a, long nlines, hipStream_t stream, int *n_partials) {
    KTimeScope kt(2, stream);
    *n_partials = (int)(nlines / LZ_FUSED);
    if (a.rc && (zw3_selected(nz, nlines) || !zw_lines_of(nz, nlines))) {
        c21hip_set_error("fused pass Z with recombinations needs the 16-lane wave kernel (z-lines of 256 / 512 points)");
        return C21CM_VALUE_ERROR;
    }
    if (zw3_selected(nz, nlines)) {  // 1024-point lines: three radix-8 stages per wave
        const float2 *twH = twiddles(nz / 2);
        const float2 *twN = twiddles(nz);
        if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
        *n_partials = (int)(nlines / (kBlock / 64));
        const dim3 grid((unsigned)(nlines / (kBlock / 64)));
        if (a.x_main)
            hipLaunchKernelGGL(zw3_ionise_kernel<true>, grid, dim3(kBlock), 0, stream, a, twH, twN);
        else
            hipLaunchKernelGGL(zw3_ionise_kernel<false>, grid, dim3(kBlock), 0, stream, a, twH, twN);
        LAUNCH_CHECK();
        return 0;
    }
    if (int zwl = zw_lines_of(nz, nlines)) {
        const float2 *twH = twiddles(nz / 2);
        const float2 *twN = twiddles(nz);
        if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
        const int zblk = (nz == 512) ? C21X_ZW_BLOCK : kBlock;  // (A = 16, P = 16)
        if (nz == 512) zwl = zblk / 16;
        *n_partials = (int)(nlines / zwl);
        const dim3 grid((unsigned)(nlines / zwl));
#define ZW_FUSED(A, TS, P) \
    hipLaunchKernelGGL((zw_ionise_kernel<A, TS, P>), grid, dim3(zblk), 0, stream, a, twH, twN)
#define ZW_FUSED_RC(A, P) \
    hipLaunchKernelGGL((zw_ionise_kernel<A, false, P, true>), grid, dim3(zblk), 0, stream, a, twH, twN)
        if (a.rc && a.x_main) {  // recombinations AND the x_e grid of a spin-temperature run (round 4)
            if (nz == 256)
                hipLaunchKernelGGL((zw_ionise_kernel<16, true, 8, true>), grid, dim3(zblk), 0, stream, a, twH, twN);
            else if (nz == 512)
                hipLaunchKernelGGL((zw_ionise_kernel<16, true, 16, true>), grid, dim3(zblk), 0, stream, a, twH, twN);
            else {
                c21hip_set_error("fused pass Z with recombinations and an x_e grid: 256- / 512-point z-lines only");
                return C21CM_VALUE_ERROR;
            }
        } else if (a.rc) {
            if (nz == 256)
                ZW_FUSED_RC(16, 8);
            else if (nz == 512)
                ZW_FUSED_RC(16, 16);
            else
                ZW_FUSED_RC(32, 16);
        } else if (a.x_main) {
            if (nz == 256)
                ZW_FUSED(16, true, 8);
            else if (nz == 512)
                ZW_FUSED(16, true, 16);
            else
                ZW_FUSED(32, true, 16);
        } else if (nz == 256)
            ZW_FUSED(16, false, 8);
        else if (nz == 512)
            ZW_FUSED(16, false, 16);
        else
            ZW_FUSED(32, false, 16);
#undef ZW_FUSED
#undef ZW_FUSED_RC
        LAUNCH_CHECK();
        return 0;
    }
    if (a.x_main) {
        c21hip_set_error("fused pass Z with an x_e grid needs 256-, 512- or 1024-point z-lines");
        return C21CM_VALUE_ERROR;
    }
    switch (nz) {
        case 64: return launch_z_fused<64>(a, nlines, stream);
        case 128: return launch_z_fused<128>(a, nlines, stream);
        case 192: return launch_z_fused<192>(a, nlines, stream);
        case 256: return launch_z_fused<256>(a, nlines, stream);
        case 384: return launch_z_fused<384>(a, nlines, stream);
        case 512: return launch_z_fused<512>(a, nlines, stream);
        case 768: return launch_z_fused<768>(a, nlines, stream);
        case 1024: return launch_z_fused<1024>(a, nlines, stream);
        case 1536: return launch_z_fused<1536>(a, nlines, stream);
        default:
            c21hip_set_error("native FFT: unsupported z length %d for the fused pass", nz);
            return C21CM_VALUE_ERROR;
    }
}

template <int NZ>
int launch_z_r2c(const ZFwdArgs &a, long nlines, hipStream_t stream) {
    constexpr int H = NZ / 2;
    const float2 *twH = twiddles(H);
    const float2 *twN = twiddles(NZ);
    if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
    const size_t lds = sizeof(float2) * ((size_t)H * (LZ_PLAIN + 1) + H + H / 2 + 1);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)z_r2c_kernel<NZ>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((z_r2c_kernel<NZ>), dim3((unsigned)(nlines / LZ_PLAIN)), dim3(kBlock), lds,
                       stream, a, twH, twN);
    LAUNCH_CHECK();
    return 0;
}

// C21CM_Z_R2C=w: 1024-point forward z-lines on the 32-values-per-lane kernel (A/B switch)
bool zw3_r2c_on() {
    const char *e = getenv("C21CM_Z_R2C");
    return !(e && e[0] == 'w');
}

int dispatch_z_r2c(int nz, const ZFwdArgs &a, long nlines, hipStream_t stream) {
    if (nz == 1024 && a.in_zstride % 2 == 0 && zw3_selected(nz, nlines) && nlines % (2 * (kBlock / 64)) == 0 &&
        zw3_r2c_on()) {
        const float2 *twH = twiddles(nz / 2);
        const float2 *twN = twiddles(nz);
        if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
        hipLaunchKernelGGL((zw3_r2c_kernel<2>), dim3((unsigned)(nlines / (2 * (kBlock / 64)))), dim3(kBlock), 0,
                           stream, a, twH, twN);
        LAUNCH_CHECK();
        return 0;
    }
    if (const int zwl = (a.in_zstride % 2 == 0) ? zw_lines_of(nz, nlines) : 0) {
        const float2 *twH = twiddles(nz / 2);
        const float2 *twN = twiddles(nz);
        if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
        const dim3 grid((unsigned)(nlines / zwl));
        if (nz == 256)
            hipLaunchKernelGGL((zw_r2c_kernel<16, 8>), grid, dim3(kBlock), 0, stream, a, twH, twN);
        else if (nz == 512)
            hipLaunchKernelGGL((zw_r2c_kernel<16, 16>), grid, dim3(kBlock), 0, stream, a, twH, twN);
        else
            hipLaunchKernelGGL((zw_r2c_kernel<32, 16>), grid, dim3(kBlock), 0, stream, a, twH, twN);
        LAUNCH_CHECK();
        return 0;
    }
    switch (nz) {
        case 64: return launch_z_r2c<64>(a, nlines, stream);
        case 128: return launch_z_r2c<128>(a, nlines, stream);
        case 192: return launch_z_r2c<192>(a, nlines, stream);
        case 256: return launch_z_r2c<256>(a, nlines, stream);
        case 384: return launch_z_r2c<384>(a, nlines, stream);
        case 512: return launch_z_r2c<512>(a, nlines, stream);
        case 768: return launch_z_r2c<768>(a, nlines, stream);
        case 1024: return launch_z_r2c<1024>(a, nlines, stream);
        case 1536: return launch_z_r2c<1536>(a, nlines, stream);
        case 2048: return launch_z_r2c<2048>(a, nlines, stream);
        default:
            c21hip_set_error("native FFT: unsupported z length %d", nz);
            return C21CM_VALUE_ERROR;
    }
}

// single-grid pass Z walks the lines backwards (C21CM_ZW_REVERSE=0: forwards): 0.268 -> 0.251 ms for the
// closed-form pass Z at 512^3, 0.243 -> 0.219 ms for the store + extrema pass
static int zw_reverse_default() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("C21CM_ZW_REVERSE");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

template <int EPI = 0>
int dispatch_z_c2r(int nz, const ZPassArgs &a_, long nlines, hipStream_t stream) {
    ZPassArgs a = a_;
    a.reverse = zw_reverse_default();  // (every caller runs this right after the pass Y of the same spectrum)
    // (partial arrays of the EPI variants are sized for 16 lines per workgroup: P = 16 only)
    if (const int zwl = (a.out_zstride % 2 == 0 && (EPI == 0 || nz != 256)) ? zw_lines_of(nz, nlines) : 0) {
        const float2 *twH = twiddles(nz / 2);
        const float2 *twN = twiddles(nz);
        if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
        const dim3 grid((unsigned)(nlines / zwl));
        if (nz == 256)
            hipLaunchKernelGGL((zw_c2r_kernel<16, EPI, 8>), grid, dim3(kBlock), 0, stream, a, twH, twN);
        else if (nz == 512)
            hipLaunchKernelGGL((zw_c2r_kernel<16, EPI, 16>), grid, dim3(kBlock), 0, stream, a, twH, twN);
        else
            hipLaunchKernelGGL((zw_c2r_kernel<32, EPI, 16>), grid, dim3(kBlock), 0, stream, a, twH, twN);
        LAUNCH_CHECK();
        return 0;
    }
    if constexpr (EPI == 6 || EPI == 7 || EPI == 9 || EPI == 10) {
        c21hip_set_error("pass Z with the deferred / banded barrier (EPI 6, 7, 9) or the extrema alone (10) needs the wave-level kernel");
        return C21CM_VALUE_ERROR;
    } else
    switch (nz) {
        case 64: return launch_z_c2r<64, EPI>(a, nlines, stream);
        case 128: return launch_z_c2r<128, EPI>(a, nlines, stream);
        case 192: return launch_z_c2r<192, EPI>(a, nlines, stream);
        case 256: return launch_z_c2r<256, EPI>(a, nlines, stream);
        case 384: return launch_z_c2r<384, EPI>(a, nlines, stream);
        case 512: return launch_z_c2r<512, EPI>(a, nlines, stream);
        case 768: return launch_z_c2r<768, EPI>(a, nlines, stream);
        case 1024: return launch_z_c2r<1024, EPI>(a, nlines, stream);
        case 1536: return launch_z_c2r<1536, EPI>(a, nlines, stream);
        case 2048: return launch_z_c2r<2048, EPI>(a, nlines, stream);
        default:
            c21hip_set_error("native FFT: unsupported z length %d", nz);
            return C21CM_VALUE_ERROR;
    }
}

void fill_filter(FilterParams &fp, int filter_type, float R, float R_param, double box_len,
                 double box_len_z) {
    fp.type = filter_type;
    {
        static int libm = -1;
        if (libm < 0) {
            const char *e = getenv("C21CM_TRIG");
            libm = (e && e[0] == 'l') ? 1 : 0;  // C21CM_TRIG=libm
        }
        fp.libm_trig = libm;
    }
    fp.R = R;
    fp.R_param = R_param;
    fp.dkx = 2.0 * M_PI / box_len;
    fp.dky = 2.0 * M_PI / box_len;
    fp.dkz = 2.0 * M_PI / box_len_z;
    fp.mfp = ExpMfpConsts{};
    if (filter_type == 3) {
        // filtering.c:320-322 (float division, double exp) and :83-94
        const double exp_term = exp((double)(-R / R_param));
        const double ratio = (double)R_param / (double)R;
        fp.mfp.R = (double)R;
        fp.mfp.ratio = ratio;
        fp.mfp.ratio2 = pow(ratio, 2);
        fp.mfp.ratio3 = pow(ratio, 3);
        fp.mfp.exp_term = exp_term;
        fp.mfp.ts_0 =
            6 * pow(ratio, 3) - exp_term * (6 * pow(ratio, 3) + 6 * pow(ratio, 2) + 3 * ratio);
        fp.mfp.ts_2 =
            exp_term * (2 * pow(ratio, 2) + 0.5 * ratio) - 2 * fp.mfp.ts_0 * pow(ratio, 2);
    }
}
}  // namespace

// nx, ny in {64..1024}, nz in {64..2048}, all powers of two
// line lengths of the native passes: 2^L (64 .. 1024; z-lines up to 2048) and 3 * 2^L for the
// reference's default DIM = 3 HII_DIM grids (192, 384, 768, 1536)
static bool native_len(int n, int pow2_max) {
    return (pow2(n) && n >= 64 && n <= pow2_max) || n == 192 || n == 384 || n == 768 || n == 1536;
}
extern "C" int c21hip_native_fft_supported(int nx, int ny, int nz) {
    return native_len(nx, 1024) && native_len(ny, 1024) && native_len(nz, 2048);
}
// two radii per pass-X sweep need two line tiles in LDS
extern "C" int c21hip_pair_sweep_supported(int nx) { return nx <= 512; }

// ---- the timing hook of this file for launches issued elsewhere (bench.py: c21hip_ktime_*)
extern "C" const void *c21hip_twiddles_dev(int n) { return twiddles(n); }
extern "C" void *c21hip_ktime_begin(int kind, void *stream) {
    return g_ktime_on ? new KTimeScope(kind, (hipStream_t)stream) : nullptr;
}
extern "C" void c21hip_ktime_end(void *scope) { delete static_cast<KTimeScope *>(scope); }


// Placement probe (ionize_driver.c: place_work_partner): pass Y of two work spectra in one launch, in place on
// whatever the buffers hold, `reps` launches timed with events on the stream -> *ms per launch.  Which physical
// region of the HBM two buffers written by one launch sit in decides 10-20 % of that launch's time
// (profiles/r05_placement_study.txt); the addresses do not tell, a timed launch does.
extern "C" int c21hip_probe_pass_y2(float *work_a, float *work_b, int nx, int ny, int nz, int reps, float *ms,
                                    void *stream_) {
    if (!c21hip_native_fft_supported(nx, ny, nz) || !work_a || !work_b || reps < 1 || !ms) return C21CM_VALUE_ERROR;
    hipStream_t stream = (hipStream_t)stream_;
    const int H = nz / 2;
    const long nlines = (long)nx * ny;
    LinePassArgs a{};
    a.fp.type = -1;
    a.n_y = ny;
    a.n_z = nz;
    a.out_scale = 1.0f;
    a.n_geo = 2;
    a.n_grids = 2;
    a.g0 = geo_y_main(nx, ny, H, line_tile_cols(ny), split_xb_log2(nx));
    a.g1 = geo_y_nyq(nx, ny, line_tile_cols(ny));
    a.g1_strided = 1;
    float2 *wk[2] = {reinterpret_cast<float2 *>(work_a), reinterpret_cast<float2 *>(work_b)};
    for (int g = 0; g < 2; g++) {
        geo_ptrs(a.g0, g, wk[g], wk[g]);
        geo_ptrs(a.g1, g, wk[g] + nlines * H, wk[g] + nlines * H);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        return C21CM_IO_ERROR;
    }
    int st = dispatch_line_pass<+1>(ny, a, 0, stream);  // warm-up (first touch of the pages)
    (void)hipEventRecord(e0, stream);
    for (int r = 0; r < reps && !st; r++) st = dispatch_line_pass<+1>(ny, a, 0, stream);
    (void)hipEventRecord(e1, stream);
    float t = 0.f;
    if (!st && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess)) {
        (void)hipGetLastError();
        st = C21CM_IO_ERROR;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / (float)reps;
    return st;
}

// Diagnostic: pass Y of ONE spectrum from `src` to `dst` (the same pointer: in place, what the product runs),
// timed like c21hip_probe_pass_y2 (tools/scratch/placement_probe5.py: is an out-of-place pass Y into another region
// of the HBM faster than the in-place one?)
extern "C" int c21hip_probe_pass_y1(const float *src, float *dst, int nx, int ny, int nz, int reps, float *ms,
                                    void *stream_) {
    if (!c21hip_native_fft_supported(nx, ny, nz) || !src || !dst || reps < 1 || !ms) return C21CM_VALUE_ERROR;
    hipStream_t stream = (hipStream_t)stream_;
    const int H = nz / 2;
    const long nlines = (long)nx * ny;
    LinePassArgs a{};
    a.fp.type = -1;
    a.n_y = ny;
    a.n_z = nz;
    a.out_scale = 1.0f;
    a.n_geo = 2;
    a.n_grids = 1;
    a.g0 = geo_y_main(nx, ny, H, line_tile_cols(ny), split_xb_log2(nx));
    a.g1 = geo_y_nyq(nx, ny, line_tile_cols(ny));
    a.g1_strided = 1;
    const float2 *s2 = reinterpret_cast<const float2 *>(src);
    float2 *d2 = reinterpret_cast<float2 *>(dst);
    geo_ptrs(a.g0, 0, s2, d2);
    geo_ptrs(a.g1, 0, s2 + nlines * H, d2 + nlines * H);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        return C21CM_IO_ERROR;
    }
    int st = dispatch_line_pass<+1>(ny, a, 0, stream);
    (void)hipEventRecord(e0, stream);
    for (int r = 0; r < reps && !st; r++) st = dispatch_line_pass<+1>(ny, a, 0, stream);
    (void)hipEventRecord(e1, stream);
    float t = 0.f;
    if (!st && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess)) {
        (void)hipGetLastError();
        st = C21CM_IO_ERROR;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / (float)reps;
    return st;
}

// ... and of TWO spectra, each from its source to its destination (diagnostic)
extern "C" int c21hip_probe_pass_y2o(const float *src_a, float *dst_a, const float *src_b, float *dst_b, int nx, int ny,
                                     int nz, int reps, float *ms, void *stream_) {
    if (!c21hip_native_fft_supported(nx, ny, nz) || !src_a || !dst_a || !src_b || !dst_b || reps < 1 || !ms)
        return C21CM_VALUE_ERROR;
    hipStream_t stream = (hipStream_t)stream_;
    const int H = nz / 2;
    const long nlines = (long)nx * ny;
    LinePassArgs a{};
    a.fp.type = -1;
    a.n_y = ny;
    a.n_z = nz;
    a.out_scale = 1.0f;
    a.n_geo = 2;
    a.n_grids = 2;
    a.g0 = geo_y_main(nx, ny, H, line_tile_cols(ny), split_xb_log2(nx));
    a.g1 = geo_y_nyq(nx, ny, line_tile_cols(ny));
    a.g1_strided = 1;
    const float2 *sp[2] = {reinterpret_cast<const float2 *>(src_a), reinterpret_cast<const float2 *>(src_b)};
    float2 *dp[2] = {reinterpret_cast<float2 *>(dst_a), reinterpret_cast<float2 *>(dst_b)};
    for (int g = 0; g < 2; g++) {
        geo_ptrs(a.g0, g, sp[g], dp[g]);
        geo_ptrs(a.g1, g, sp[g] + nlines * H, dp[g] + nlines * H);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        return C21CM_IO_ERROR;
    }
    int st = dispatch_line_pass<+1>(ny, a, 0, stream);
    (void)hipEventRecord(e0, stream);
    for (int r = 0; r < reps && !st; r++) st = dispatch_line_pass<+1>(ny, a, 0, stream);
    (void)hipEventRecord(e1, stream);
    float t = 0.f;
    if (!st && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess)) {
        (void)hipGetLastError();
        st = C21CM_IO_ERROR;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / (float)reps;
    return st;
}

// in-loop kernel timing, see KTimeScope
extern "C" void c21hip_ktime_enable(int on) {
    for (auto &r : g_ktime) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_ktime.clear();
    g_ktime_on = on ? 1 : 0;
}
// total device time (ms) and launch count of kind `kind` since c21hip_ktime_enable(1); synchronises
extern "C" int c21hip_ktime_report(int kind, double *ms_total, int *count) {
    double t = 0.;
    int n = 0;
    for (auto &r : g_ktime) {
        if (r.kind != kind) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) return C21CM_IO_ERROR;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return C21CM_IO_ERROR;
        t += ms;
        n++;
    }
    if (ms_total) *ms_total = t;
    if (count) *count = n;
    return 0;
}

extern "C" size_t c21hip_split_floats(int nx, int ny, int nz) {
    return 2 * ((size_t)nx * ny * (size_t)(nz / 2) + (size_t)nx * ny);
}

extern "C" int c21hip_padded_to_split(const float *padded_c, float *split, int nx, int ny, int nz,
                                      void *stream) {
    const long nlines = (long)nx * ny;
    const int H = nz / 2;
    float2 *main = reinterpret_cast<float2 *>(split);
    float2 *nyq = main + nlines * H;
    size_t blocks = ((size_t)nlines * (H + 1) + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(padded_to_split_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, reinterpret_cast<const float2 *>(padded_c), main, nyq,
                       nlines, H, ny, split_xb_log2(nx));
    LAUNCH_CHECK();
    return 0;
}

// [W(kR) x] pass X (src -> work) and pass Y (work, in place) of the inverse transform of one
// or two split spectra.  src == work is allowed.  Grid g is filtered with window
// (filter_type[g], R, R_param[g]).
// The per-radius window tables of a one- or two-grid sweep in buffer `table_slot` (0..3);
// build = launch the kernel that fills them.
struct WinTables {
    const wtab_t *main[2], *nyq[2];
    int dual;
};
static int win_tables(int table_slot, int n_grids, const int filter_type[2], float R,
                      const float R_param[2], float R_star, int nx, int ny, int nz, double box_len,
                      double box_len_z, bool build, hipStream_t stream, WinTables &w) {
    static const int ws_slot[4] = {49, 46, 94, 95};
    if (table_slot < 0 || table_slot > 3) return C21CM_VALUE_ERROR;
    const int H = nz / 2;
    // the window's parameters beyond (type, R): R_param matters for types 3 and 4 only
    const bool dual = n_grids == 2 && (filter_type[0] != filter_type[1] ||
                                       (filter_type[0] >= 3 && R_param[0] != R_param[1]));
    const size_t n_main = (size_t)(ny / 2 + 1) * (nx / 2 + 1) * H;
    const size_t n_nyq = (size_t)(nx / 2 + 1) * (ny / 2 + 1);
    const size_t per = (n_main + n_nyq + 3) & ~(size_t)3;  // keep table b 16-byte aligned
    wtab_t *tab = (wtab_t *)c21hip_ws(ws_slot[table_slot], sizeof(wtab_t) * per * (dual ? 2 : 1));
    if (!tab) return C21CM_MEMORY_ALLOC_ERROR;
    WTableArgs t{};
    fill_filter(t.pa, filter_type[0], R, R_param[0], box_len, box_len_z);
    t.dual = dual ? 1 : 0;
    if (dual) fill_filter(t.pb, filter_type[1], R, R_param[1], box_len, box_len_z);
    t.nx = nx;
    t.ny = ny;
    t.nz = nz;
    t.main_a = tab;
    t.nyq_a = tab + n_main;
    t.main_b = dual ? tab + per : tab;
    t.nyq_b = t.main_b + n_main;
    if (build) {
        const size_t n_threads = n_main / 4 + n_nyq;
        const bool ms = filter_type[0] == 5 || (dual && filter_type[1] == 5);
        if (ms) {
            if (filter_type[0] == 5) ms_fill(t.ms_a, R, R_param[0], R_star);
            if (dual && filter_type[1] == 5) ms_fill(t.ms_b, R, R_param[1], R_star);
        }
        const dim3 tgrid((unsigned)((n_threads + kBlock - 1) / kBlock));
        if (ms)
            hipLaunchKernelGGL(window_table_kernel<true>, tgrid, dim3(kBlock), 0, stream, t);
        else
            hipLaunchKernelGGL(window_table_kernel<false>, tgrid, dim3(kBlock), 0, stream, t);
        LAUNCH_CHECK();
    }
    w.main[0] = t.main_a;
    w.nyq[0] = t.nyq_a;
    w.main[1] = t.main_b;
    w.nyq[1] = t.nyq_b;
    w.dual = t.dual;
    return 0;
}

// ---- the prepared node-table set of one excursion-set call (c21hip_wev_prepare)
struct WevSet {
    bool active = false;
    int filter[2] = {-1, -1};
    float R_param[2] = {0.f, 0.f};
    int nx = 0, ny = 0, nz = 0;
    double box_len = 0, box_len_z = 0;
    std::vector<float> R;
    float *nodes = nullptr;  // device: [n_tabs][n_nodes][3]
    int n_nodes = 0, n_tabs = 0;
    bool covers = true;      // the node tables reach the largest kR of the set (else: direct beyond)
    int first_type = 0;      // type of table 0
    // table of (window w, radius index r); -1: evaluated directly (sharp-k)
    int table(int w, int r) const {
        const int t = filter[w];
        if (t == 1) return -1;
        if (t == 0) return 0;
        // type 3: tables follow the top-hat table (if any); window b's follow window a's
        int base = (first_type == 0) ? 1 : 0;
        if (w == 1 && filter[0] == 3) base += (int)R.size();
        return base + r;
    }
    int find(float radius) const {
        for (size_t i = 0; i < R.size(); i++)
            if (R[i] == radius) return (int)i;
        return -1;
    }
};
WevSet g_wev;

static bool wev_type_ok(int t) { return t == 0 || t == 1 || t == 3; }
// LDS of a pass-X launch with evaluated windows (bytes)
static size_t wev_lds(int n, bool pair, int n_tabs, int n_nodes) {
    return sizeof(float2) * ((size_t)(pair ? 2 : 1) * n * TZ + n + (n >= 1024 ? n / 2 : 0)) +
           sizeof(float) * 3 * (size_t)n_nodes * n_tabs;
}
// Fill the evaluated-window members of `a` for radii R (and R2 when pair) of the active set;
// false: this launch cannot use them (the caller streams tables instead).
static bool wev_use(LinePassArgs &a, int n_grids, const int filter_type[2], const float R_param[2],
                    float R, float R2, bool pair, int nx, int ny, int nz, double box_len,
                    double box_len_z) {
    const WevSet &w = g_wev;
    if (!w.active || w.nx != nx || w.ny != ny || w.nz != nz || w.box_len != box_len ||
        w.box_len_z != box_len_z)
        return false;
    // a launch may carry both windows, or ONE grid under window a (the x_e grid of a
    // spin-temperature run) or under window b (whalo_sfr of a recombination run)
    int win0 = 0;
    auto matches = [&](int g, int win) {
        return filter_type[g] == w.filter[win] && (filter_type[g] != 3 || R_param[g] == w.R_param[win]);
    };
    if (n_grids == 2) {
        if (!matches(0, 0) || !matches(1, 1)) return false;
    } else if (!matches(0, 0)) {
        if (!matches(0, 1)) return false;
        win0 = 1;
    }
    const int ra = w.find(R), rb = pair ? w.find(R2) : ra;
    if (ra < 0 || rb < 0) return false;
    // distinct tables of this launch -> staged LDS slots
    int ids[4], n_ids = 0;
    auto slot = [&](int id) {
        if (id < 0) return -1;
        for (int i = 0; i < n_ids; i++)
            if (ids[i] == id) return i;
        ids[n_ids] = id;
        return n_ids++;
    };
    const int rr[2] = {ra, rb};
    for (int m = 0; m < 2; m++)
        for (int win = 0; win < 2; win++)
            a.wev_tab[m][win] =
                (win < n_grids && (m == 0 || pair)) ? slot(w.table(win == 0 ? win0 : win, rr[m])) : -1;
    if (n_ids > 3) return false;
    if (wev_lds(nx, pair, n_ids, w.n_nodes) > 160 * 1024) return false;
    for (int i = 0; i < 3; i++)
        a.wev_src[i] = w.nodes + 3 * (size_t)w.n_nodes * (i < n_ids ? ids[i] : ids[0 < n_ids ? 0 : 0]);
    a.wev_n_tabs = n_ids > 0 ? n_ids : 1;
    if (n_ids == 0) a.wev_src[0] = a.wev_src[1] = a.wev_src[2] = w.nodes;  // sharp-k only
    a.wev_n_nodes = w.n_nodes;
    a.wev_type[0] = filter_type[0];
    a.wev_type[1] = n_grids == 2 ? filter_type[1] : filter_type[0];
    a.wev_R[0] = R;
    a.wev_R[1] = pair ? R2 : R;
    for (int m = 0; m < 2; m++) {  // the exp-MFP window's constants per sweep member (filtering.c:320-322)
        const int win = (filter_type[0] == 3) ? 0 : ((n_grids == 2 && filter_type[1] == 3) ? 1 : -1);
        if (win < 0) break;
        const float Rm = m ? a.wev_R[1] : a.wev_R[0];
        const double ratio = (double)R_param[win] / (double)Rm;
        a.wev_mfp[m][0] = (float)ratio;
        a.wev_mfp[m][1] = (float)(ratio * ratio);
        a.wev_mfp[m][2] = (float)(ratio * ratio * ratio);
        a.wev_mfp[m][3] = (float)exp((double)(-Rm / R_param[win]));
    }
    a.wev_dkx = 2.0 * M_PI / box_len;
    a.wev_dky = 2.0 * M_PI / box_len;
    a.wev_dkz = 2.0 * M_PI / box_len_z;
    a.dual = (n_grids == 2 && (filter_type[0] != filter_type[1] ||
                               (filter_type[0] >= 3 && R_param[0] != R_param[1])))
                 ? 1 : 0;
    return true;
}

static int check_filter_request(int n_grids, int nx, int ny, int nz, const int filter_type[2],
                                int apply) {
    if (!c21hip_native_fft_supported(nx, ny, nz)) {
        c21hip_set_error("native FFT does not support %dx%dx%d", nx, ny, nz);
        return C21CM_VALUE_ERROR;
    }
    for (int g = 0; g < n_grids; g++)
        if (apply && (filter_type[g] < 0 || filter_type[g] > 5)) {
            c21hip_set_error("filter type %d is not implemented on the device", filter_type[g]);
            return C21CM_VALUE_ERROR;
        }
    return 0;
}

static int filter_xy(const float *const split_src[2], float *const split_work[2], int n_grids,
                     int nx, int ny, int nz, double box_len, double box_len_z,
                     const int filter_type[2], float R, const float R_param[2], int apply,
                     void *stream_, int phases = 7, int table_slot = 0, float R_star = 0.f) {
    // phases: 1 window table, 2 pass X, 4 pass Y (the timing hook runs them one at a time; the
    // excursion-set driver builds the tables of the next radius on a second stream).
    // table_slot: which of the table buffers this radius uses.
    int st = check_filter_request(n_grids, nx, ny, nz, filter_type, apply);
    if (st) return st;
    hipStream_t stream = (hipStream_t)stream_;
    const int H = nz / 2;
    const long nlines = (long)nx * ny;
    LinePassArgs a{};
    fill_filter(a.fp, apply ? filter_type[0] : -1, R, R_param[0], box_len, box_len_z);
    a.n_y = ny;
    a.n_z = nz;
    a.out_scale = 1.0f;
    int fmode = apply ? 3 : 0;
    if (fmode == 3 && filter_type[0] != 5 && wev_use(a, n_grids, filter_type, R_param, R, R, false, nx,
                                                      ny, nz, box_len, box_len_z))
        fmode = g_wev.covers ? 6 : 8;  // windows evaluated in pass X: no table is built or read
    if (fmode == 3) {
        WinTables w;
        if ((st = win_tables(table_slot, n_grids, filter_type, R, R_param, R_star, nx, ny, nz,
                             box_len, box_len_z, (phases & 1) != 0, stream, w)))
            return st;
        for (int i = 0; i < 2; i++) {
            a.wt_main[i] = w.main[i];
            a.wt_nyq[i] = w.nyq[i];
        }
        a.dual = w.dual;
    }
    if (!(phases & 6)) return 0;
    // ---- pass X: main block + Nyquist plane (x grids) in one launch
    a.n_geo = 2;
    a.n_grids = n_grids;
    a.g0 = geo_x_main(ny, H, line_tile_cols(nx), split_xb_log2(nx));
    a.g1 = geo_x_nyq(ny, line_tile_cols(nx));
    for (int g = 0; g < n_grids; g++) {
        const float2 *src = reinterpret_cast<const float2 *>(split_src[g]);
        float2 *work = reinterpret_cast<float2 *>(split_work[g]);
        geo_ptrs(a.g0, g, src, work);
        geo_ptrs(a.g1, g, src + nlines * H, work + nlines * H);
    }
    if ((phases & 2) && (st = dispatch_line_pass<+1>(nx, a, fmode, stream))) return st;
    if (!(phases & 4)) return 0;
    // ---- pass Y (in place)
    a.fp.type = -1;
    a.g0 = geo_y_main(nx, ny, H, line_tile_cols(ny), split_xb_log2(nx));
    a.g1 = geo_y_nyq(nx, ny, line_tile_cols(ny));
    a.g1_strided = 1;
    for (int g = 0; g < n_grids; g++) {
        float2 *work = reinterpret_cast<float2 *>(split_work[g]);
        geo_ptrs(a.g0, g, work, work);
        geo_ptrs(a.g1, g, work + nlines * H, work + nlines * H);
    }
    return dispatch_line_pass<+1>(ny, a, 0, stream);
}

extern "C" int c21hip_split_filter_xy(const float *split_src, float *split_work, int nx, int ny,
                                      int nz, double box_len, double box_len_z, int filter_type,
                                      float R, float R_param, int apply, void *stream_) {
    const float *src[2] = {split_src, nullptr};
    float *work[2] = {split_work, nullptr};
    const int ft[2] = {filter_type, 0};
    const float rp[2] = {R_param, 0.f};
    return filter_xy(src, work, 1, nx, ny, nz, box_len, box_len_z, ft, R, rp, apply, stream_);
}

// Two grids of the same shape in one sweep (the density and emissivity spectra of the
// excursion-set loop), each with its own window of the same radius.  tables_ready != 0: the
// window tables were already built into buffer `table_slot` by c21hip_window_tables.
extern "C" int c21hip_split_filter_xy2(const float *src_a, float *work_a, int filter_a,
                                       float R_param_a, const float *src_b, float *work_b,
                                       int filter_b, float R_param_b, int nx, int ny, int nz,
                                       double box_len, double box_len_z, float R, int apply,
                                       int table_slot, int tables_ready, void *stream_) {
    const float *src[2] = {src_a, src_b};
    float *work[2] = {work_a, work_b};
    const int ft[2] = {filter_a, filter_b};
    const float rp[2] = {R_param_a, R_param_b};
    return filter_xy(src, work, 2, nx, ny, nz, box_len, box_len_z, ft, R, rp, apply, stream_,
                     tables_ready ? 6 : 7, table_slot);
}

// The same for TWO radii R and R2 in one pass-X sweep (FMODE 5: every source tile is read
// once and transformed twice), followed by a pass Y per radius: work_*/table_slot belong to R,
// work_*2/table_slot2 to R2.  Lines of 1024 points keep one register set and are not served.
static int filter_xy_pair(const float *src_a, float *work_a, float *work_a2, int filter_a,
                          float R_param_a, const float *src_b, float *work_b, float *work_b2,
                          int filter_b, float R_param_b, int nx, int ny, int nz, double box_len,
                          double box_len_z, float R, float R2, int table_slot, int table_slot2,
                          int phases, void *stream_, int n_grids = 2, bool must_eval = false) {
    // phases: 1 window tables, 2 pass X, 4 pass Y of the first radius, 8 pass Y of the second
    // n_grids = 1: grid a only, with window a of tables built for a two-grid sweep (phases & 1
    // must be clear)
    const int tables_ready = !(phases & 1);
    const int ft[2] = {filter_a, filter_b};
    const float rp[2] = {R_param_a, R_param_b};
    int st = check_filter_request(n_grids, nx, ny, nz, ft, 1);
    if (st) return st;
    if (nx >= 1024 || table_slot == table_slot2 || (n_grids != 2 && !tables_ready)) {
        c21hip_set_error("two-radius sweep: nx = %d / table slots %d, %d not supported", nx,
                         table_slot, table_slot2);
        return C21CM_VALUE_ERROR;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const int H = nz / 2;
    const long nlines = (long)nx * ny;
    LinePassArgs a{};
    a.fp.type = -1;
    a.n_y = ny;
    a.n_z = nz;
    a.out_scale = 1.0f;
    // (must_eval: the one-grid sweep of the Eulerian loops -- no tables were built for it; its pass-Y-only calls
    //  need no window at all)
    const bool evaluated = (must_eval && !(phases & 2))
                               ? true
                               : wev_use(a, n_grids, ft, rp, R, R2, true, nx, ny, nz, box_len, box_len_z);
    if (must_eval && !evaluated) {
        c21hip_set_error("two-radius sweep of one grid: the evaluated windows do not cover radii %g, %g", (double)R,
                         (double)R2);
        return C21CM_VALUE_ERROR;
    }
    if (!evaluated) {
        WinTables w, w2;
        if ((st = win_tables(table_slot, n_grids, ft, R, rp, 0.f, nx, ny, nz, box_len, box_len_z,
                             !tables_ready, stream, w)))
            return st;
        if ((st = win_tables(table_slot2, n_grids, ft, R2, rp, 0.f, nx, ny, nz, box_len, box_len_z,
                             !tables_ready, stream, w2)))
            return st;
        for (int i = 0; i < 2; i++) {
            a.wt_main[i] = w.main[i];
            a.wt_nyq[i] = w.nyq[i];
            a.wt2_main[i] = w2.main[i];
            a.wt2_nyq[i] = w2.nyq[i];
        }
        a.dual = w.dual;
    }
    a.n_geo = 2;
    a.n_grids = n_grids;
    a.g0 = geo_x_main(ny, H, line_tile_cols(nx), split_xb_log2(nx));
    a.g1 = geo_x_nyq(ny, line_tile_cols(nx));
    const float2 *src[2] = {reinterpret_cast<const float2 *>(src_a),
                            reinterpret_cast<const float2 *>(src_b)};
    float2 *work[2] = {reinterpret_cast<float2 *>(work_a), reinterpret_cast<float2 *>(work_b)};
    float2 *work2[2] = {reinterpret_cast<float2 *>(work_a2), reinterpret_cast<float2 *>(work_b2)};
    for (int g = 0; g < n_grids && (phases & 2); g++) {
        geo_ptrs(a.g0, g, src[g], work[g]);
        geo_ptrs(a.g1, g, src[g] + nlines * H, work[g] + nlines * H);
        a.g0.dst2[g] = work2[g];
        a.g1.dst2[g] = work2[g] + nlines * H;
    }
    if ((phases & 2) &&
        (st = dispatch_line_pass<+1>(nx, a, evaluated ? (g_wev.covers ? 7 : 9) : 5, stream)))
        return st;
    if (!(phases & 12)) return 0;
    // ---- pass Y (in place), one launch per radius
    for (int r = 0; r < 2; r++) {
        if (!(phases & (4 << r))) continue;
        float2 *const *wk = r ? work2 : work;
        a.g0 = geo_y_main(nx, ny, H, line_tile_cols(ny), split_xb_log2(nx));
        a.g1 = geo_y_nyq(nx, ny, line_tile_cols(ny));
        a.g1_strided = 1;
        for (int g = 0; g < n_grids; g++) {
            geo_ptrs(a.g0, g, wk[g], wk[g]);
            geo_ptrs(a.g1, g, wk[g] + nlines * H, wk[g] + nlines * H);
        }
        if ((st = dispatch_line_pass<+1>(ny, a, 0, stream))) return st;
    }
    return 0;
}

extern "C" int c21hip_split_filter_xy2_pair(const float *src_a, float *work_a, float *work_a2,
                                            int filter_a, float R_param_a, const float *src_b,
                                            float *work_b, float *work_b2, int filter_b,
                                            float R_param_b, int nx, int ny, int nz,
                                            double box_len, double box_len_z, float R, float R2,
                                            int table_slot, int table_slot2, int phases,
                                            void *stream_) {
    // phases: 1 build the window tables, 2 pass X, 4 / 8 pass Y of the first / second radius
    return filter_xy_pair(src_a, work_a, work_a2, filter_a, R_param_a, src_b, work_b, work_b2,
                          filter_b, R_param_b, nx, ny, nz, box_len, box_len_z, R, R2, table_slot,
                          table_slot2, phases, stream_);
}

// ONE grid, two radii per sweep, with window a of the tables in table_slot / table_slot2 (built
// for the two-grid sweep of the same radii): the x_e grid of a spin-temperature run.
extern "C" int c21hip_split_filter_xy_shared_pair(const float *src, float *work, float *work2,
                                                  int filter_type, int nx, int ny, int nz,
                                                  double box_len, double box_len_z, float R,
                                                  float R2, int table_slot, int table_slot2,
                                                  int phases, void *stream_) {
    return filter_xy_pair(src, work, work2, filter_type, 0.f, nullptr, nullptr, nullptr, 0, 0.f,
                          nx, ny, nz, box_len, box_len_z, R, R2, table_slot, table_slot2, phases,
                          stream_, 1);
}

// ONE grid, two radii per sweep, windows EVALUATED in the kernel (c21hip_wev_prepare with pair = 1): the
// filtered density of the Eulerian loops (round 6) -- pass X reads the unfiltered spectrum once for two radii
// (3 S instead of 4 S).  phases: 2 pass X into work / work2, 4 pass Y of `work`, 8 pass Y of `work2`; a call
// without 2 needs no source and no window (the parked second spectrum's pass Y, one radius later).
// C21CM_VALUE_ERROR (with the error text set) when the active set does not cover the launch.
extern "C" int c21hip_split_filter_x_pair1(const float *src, float *work, float *work2, int filter_type,
                                           int nx, int ny, int nz, double box_len, double box_len_z, float R,
                                           float R2, int phases, void *stream_) {
    return filter_xy_pair(src, work, work2, filter_type, 0.f, nullptr, nullptr, nullptr, 0, 0.f, nx, ny, nz,
                          box_len, box_len_z, R, R2, 0, 1, phases & ~1, stream_, 1, true);
}

// ... and TWO grids under their evaluated windows (delta and x_e of the Eulerian table loop of a spin-temperature
// run): c21hip_split_filter_xy2_pair without tables -- an error instead of a silent table lookup when the
// evaluated set does not cover the launch; calls without phase 2 run pass Y alone on (work_a, work_b) [4] or
// (work_a2, work_b2) [8].
extern "C" int c21hip_split_filter_xy2_pair_eval(const float *src_a, float *work_a, float *work_a2, int filter_a,
                                                 const float *src_b, float *work_b, float *work_b2, int filter_b,
                                                 int nx, int ny, int nz, double box_len, double box_len_z, float R,
                                                 float R2, int phases, void *stream_) {
    return filter_xy_pair(src_a, work_a, work_a2, filter_a, 0.f, src_b, work_b, work_b2, filter_b, 0.f, nx, ny, nz,
                          box_len, box_len_z, R, R2, 0, 1, phases & ~1, stream_, 2, true);
}

// One or two grids of one shell of the spin-temperature filters: windows 4 (spherical shell)
// or 5 (multiple scattering) between R_inner and R_outer (SpinTemperatureBox.c:698-700).
extern "C" int c21hip_split_filter_shell(const float *src_a, float *work_a, int filter_a,
                                         const float *src_b, float *work_b, int filter_b,
                                         int n_grids, int nx, int ny, int nz, double box_len,
                                         double box_len_z, float R_inner, float R_outer,
                                         float R_star, int apply, void *stream_) {
    const float *src[2] = {src_a, src_b};
    float *work[2] = {work_a, work_b};
    const int ft[2] = {filter_a, filter_b};
    const float rp[2] = {R_outer, R_outer};
    return filter_xy(src, work, n_grids, nx, ny, nz, box_len, box_len_z, ft, R_inner, rp, apply,
                     stream_, 7, 0, R_star);
}

// Passes X and Y of the inverse transform of op(P), P = spectrum / k^2 in the split layout:
// (axis0, axis1 < 0) -> i k_axis0 P (gradient); (axis0, axis1) -> -k_axis0 k_axis1 P.  The
// operator is applied inside pass X (FMODE 4).  InitialConditions.c:240-297.
extern "C" int c21hip_split_sepop_xy(const float *split_src, float *split_work, int nx, int ny,
                                     int nz, double box_len, double box_len_z, int axis0,
                                     int axis1, void *stream_) {
    if (!c21hip_native_fft_supported(nx, ny, nz)) {
        c21hip_set_error("native FFT does not support %dx%dx%d", nx, ny, nz);
        return C21CM_VALUE_ERROR;
    }
    if (axis0 < 0 || axis0 > 2 || axis1 > 2) return C21CM_VALUE_ERROR;
    hipStream_t stream = (hipStream_t)stream_;
    const int H = nz / 2;
    const long nlines = (long)nx * ny;
    LinePassArgs a{};
    a.fp.type = -1;
    a.n_y = ny;
    a.n_z = nz;
    a.out_scale = 1.0f;
    int e[3] = {0, 0, 0};
    e[axis0]++;
    if (axis1 >= 0) e[axis1]++;
    a.op_ex = e[0];
    a.op_ey = e[1];
    a.op_ez = e[2];
    a.op_imag = axis1 < 0;
    a.op_sign = axis1 < 0 ? 1. : -1.;
    a.op_dkx = 2.0 * M_PI / box_len;
    a.op_dky = 2.0 * M_PI / box_len;
    a.op_dkz = 2.0 * M_PI / box_len_z;
    a.n_geo = 2;
    a.n_grids = 1;
    a.g0 = geo_x_main(ny, H, line_tile_cols(nx), split_xb_log2(nx));
    a.g1 = geo_x_nyq(ny, line_tile_cols(nx));
    const float2 *src = reinterpret_cast<const float2 *>(split_src);
    float2 *work = reinterpret_cast<float2 *>(split_work);
    geo_ptrs(a.g0, 0, src, work);
    geo_ptrs(a.g1, 0, src + nlines * H, work + nlines * H);
    int st = dispatch_line_pass<+1>(nx, a, 4, stream);
    if (st) return st;
    a.g0 = geo_y_main(nx, ny, H, line_tile_cols(ny), split_xb_log2(nx));
    a.g1 = geo_y_nyq(nx, ny, line_tile_cols(ny));
    a.g1_strided = 1;
    geo_ptrs(a.g0, 0, work, work);
    geo_ptrs(a.g1, 0, work + nlines * H, work + nlines * H);
    return dispatch_line_pass<+1>(ny, a, 0, stream);
}

// The window tables of one radius for the two-grid sweep, into buffer `table_slot`, on
// `stream_` (any stream: they depend on no grid data).
extern "C" int c21hip_window_tables(int table_slot, int filter_a, float R_param_a, int filter_b,
                                    float R_param_b, int nx, int ny, int nz, double box_len,
                                    double box_len_z, float R, void *stream_) {
    const float *src[2] = {nullptr, nullptr};
    float *work[2] = {nullptr, nullptr};
    const int ft[2] = {filter_a, filter_b};
    const float rp[2] = {R_param_a, R_param_b};
    return filter_xy(src, work, 2, nx, ny, nz, box_len, box_len_z, ft, R, rp, 1, stream_, 1,
                     table_slot);
}

// Prepare the node tables of W(x) for the radii R[0..n_R) of one excursion-set call (windows a / b
// as in c21hip_split_filter_xy2).  *enabled = 1: passes X of these radii evaluate their windows
// in the kernel and no 3-D window table is built or read (c21hip_window_tables becomes a no-op
// the caller can skip); 0: not applicable here (filter types other than top-hat / sharp-k /
// exp-MFP, line lengths without the kernel, node tables that do not fit in LDS beside the tiles,
// or C21CM_WINDOWS=table) and the table path stays.  c21hip_wev_release() ends the set.
// The static half of that decision -- filter types, line lengths, the number of node tables a
// launch stages and the LDS left for them beside the tiles -- shared by c21hip_wev_prepare and
// c21hip_wev_applicable so that "applicable" can never promise what "prepare" then declines
// (ADVICE r3).  *cap = nodes per table that fit; *worst = tables of the largest launch.
static bool wev_static_ok(int filter_a, int filter_b, int n_grids, int nx, int ny, int nz, int pair,
                          int *worst_out, int *cap_out) {
    if (!wev_type_ok(filter_a) || (n_grids == 2 && !wev_type_ok(filter_b))) return false;
    if (nx < 128 || (nx & (nx - 1)) || nx > 1024 || !c21hip_native_fft_supported(nx, ny, nz))
        return false;
    const int f1 = n_grids == 2 ? filter_b : filter_a;
    const bool any_tophat = (filter_a == 0) || (n_grids == 2 && f1 == 0);
    const int n_mfp_windows = (filter_a == 3 ? 1 : 0) + (n_grids == 2 && f1 == 3 ? 1 : 0);
    // the largest launch: one top-hat table + one exp-MFP table per window and sweep member
    const int worst = (any_tophat ? 1 : 0) + n_mfp_windows * (pair && nx <= 512 ? 2 : 1);
    if (worst > 3 || n_mfp_windows > 1) return false;
    // node tables as long as the LDS beside the tiles allows (the windows beyond their range are
    // evaluated directly); at least x = 32
    const size_t fixed = wev_lds(nx, pair && nx <= 512, 0, 0);
    const size_t room = fixed < 158 * 1024 ? 158 * 1024 - fixed : 0;
    const int cap = (int)(room / (sizeof(float) * 3 * (size_t)(worst > 0 ? worst : 1)));
    if (cap < 32 * 4 + 3) return false;
    if (worst_out) *worst_out = worst;
    if (cap_out) *cap_out = cap;
    return true;
}

extern "C" int c21hip_wev_prepare(int filter_a, float R_param_a, int filter_b, float R_param_b,
                                  int n_grids, const float *R, int n_R, int nx, int ny, int nz,
                                  double box_len, double box_len_z, int pair, int *enabled,
                                  void *stream_) {
    *enabled = 0;
    g_wev.active = false;
    const char *env = getenv("C21CM_WINDOWS");  // read per call: tests switch it
    const int by_table = (env && env[0] == 't') ? 1 : 0;
    // (1024-point lines kept the tables until the fused first / last stages freed their registers:
    // the evaluating kernel now spills 5 VGPRs, the table-streaming one 111 -- 3.9 against 4.65 ms per
    // pass X at 1024^3, 417 against 461-484 ms per call; C21CM_WINDOWS=table keeps the tables)
    int worst = 0, cap = 0;
    if (by_table || n_R < 1 || !wev_static_ok(filter_a, filter_b, n_grids, nx, ny, nz, pair, &worst, &cap))
        return 0;
    WevSet &w = g_wev;
    w.filter[0] = filter_a;
    w.filter[1] = n_grids == 2 ? filter_b : filter_a;
    w.R_param[0] = R_param_a;
    w.R_param[1] = n_grids == 2 ? R_param_b : R_param_a;
    w.nx = nx, w.ny = ny, w.nz = nz;
    w.box_len = box_len, w.box_len_z = box_len_z;
    w.R.assign(R, R + n_R);
    float R_max = 0.f;
    for (int i = 0; i < n_R; i++) R_max = R[i] > R_max ? R[i] : R_max;
    const double kx = M_PI * nx / box_len, ky = M_PI * ny / box_len, kz = M_PI * nz / box_len_z;
    const double x_max = sqrt(kx * kx + ky * ky + kz * kz) * R_max * (1. + 1e-6);
    w.n_nodes = (int)(x_max * 4.) + 3;
    const bool any_tophat = (w.filter[0] == 0) || (n_grids == 2 && w.filter[1] == 0);
    const int n_mfp_windows = (w.filter[0] == 3 ? 1 : 0) + (n_grids == 2 && w.filter[1] == 3 ? 1 : 0);
    w.first_type = any_tophat ? 0 : 3;
    w.n_tabs = (any_tophat ? 1 : 0) + n_mfp_windows * n_R;
    {
        const int need = w.n_nodes;
        if (w.n_nodes > cap) w.n_nodes = cap;
        if (w.n_nodes > 4096) w.n_nodes = 4096;
        w.covers = (w.n_nodes == need);
    }
    if (w.n_tabs == 0) {  // sharp-k only: nothing to tabulate, one dummy node table
        w.n_tabs = 1;
        w.first_type = 0;
    }
    hipStream_t stream = (hipStream_t)stream_;
    w.nodes = (float *)c21hip_ws(240, sizeof(float) * 3 * (size_t)w.n_nodes * w.n_tabs);
    const int n_mfp = n_mfp_windows * n_R;
    ExpMfpConsts *mfp_dev = (ExpMfpConsts *)c21hip_ws(241, sizeof(ExpMfpConsts) * (size_t)(n_mfp > 0 ? n_mfp : 1));
    if (!w.nodes || !mfp_dev) return C21CM_MEMORY_ALLOC_ERROR;
    if (n_mfp > 0) {
        static std::vector<ExpMfpConsts> host;  // must outlive the asynchronous copy
        host.resize(n_mfp);
        int o = 0;
        for (int win = 0; win < (n_grids == 2 ? 2 : 1); win++) {
            if (w.filter[win] != 3) continue;
            for (int i = 0; i < n_R; i++) {
                FilterParams fp;
                fill_filter(fp, 3, R[i], w.R_param[win], box_len, box_len_z);
                host[o++] = fp.mfp;
            }
        }
        if (hipMemcpyAsync(mfp_dev, host.data(), sizeof(ExpMfpConsts) * n_mfp, hipMemcpyHostToDevice,
                           stream) != hipSuccess)
            return C21CM_IO_ERROR;
        if (hipStreamSynchronize(stream) != hipSuccess) return C21CM_IO_ERROR;
    }
    WNodeArgs a{};
    a.out = w.nodes;
    a.n_nodes = w.n_nodes;
    a.n_tabs = w.n_tabs;
    a.first_type = w.first_type;
    a.mfp = mfp_dev;
    const int total = w.n_nodes * w.n_tabs;
    hipLaunchKernelGGL(window_nodes_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, stream, a);
    LAUNCH_CHECK();
    w.active = true;
    *enabled = 1;
    return 0;
}
extern "C" void c21hip_wev_release(void) { g_wev.active = false; }
// 1: c21hip_wev_prepare would enable evaluated windows for these filters on this grid
extern "C" int c21hip_wev_applicable(int filter_a, int filter_b, int n_grids, int nx, int ny, int nz) {
    const char *env = getenv("C21CM_WINDOWS");
    if (env && env[0] == 't') return 0;
    // with the two-radius sweep assumed (the larger LDS footprint): whatever this accepts,
    // c21hip_wev_prepare accepts with pair = 0 or 1
    return wev_static_ok(filter_a, filter_b, n_grids, nx, ny, nz, 1, nullptr, nullptr) ? 1 : 0;
}
// ONE grid, two radii per sweep, under a window of the prepared set (a or b: matched by type
// and parameter); evaluated windows only
extern "C" int c21hip_split_filter_xy_single_pair(const float *src, float *work, float *work2,
                                                  int filter_type, float R_param, int nx, int ny,
                                                  int nz, double box_len, double box_len_z, float R,
                                                  float R2, int phases, void *stream_) {
    if (!g_wev.active) {
        c21hip_set_error("single-grid pair sweep: needs a prepared window set (c21hip_wev_prepare)");
        return C21CM_VALUE_ERROR;
    }
    return filter_xy_pair(src, work, work2, filter_type, R_param, nullptr, nullptr, nullptr, 0, 0.f,
                          nx, ny, nz, box_len, box_len_z, R, R2, 0, 1, phases & ~1, stream_, 1);
}

// Forward transform into the split layout: real rows (in_zstride floats, scale-and-clip on
// load; pass lo > hi to disable the clip) -> pass Z r2c -> pass Y -> pass X, the result
// multiplied by out_scale (1/N for prepare_box_for_filtering, exact for power-of-two N).
extern "C" int c21hip_split_r2c(const float *real_in, long in_zstride, float *split_out, int nx,
                                int ny, int nz, double factor, double lo, double hi,
                                float out_scale, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!c21hip_native_fft_supported(nx, ny, nz)) {
        c21hip_set_error("native FFT does not support %dx%dx%d", nx, ny, nz);
        return C21CM_VALUE_ERROR;
    }
    const int H = nz / 2;
    const long nlines = (long)nx * ny;
    float2 *o_main = reinterpret_cast<float2 *>(split_out);
    float2 *o_nyq = o_main + nlines * H;
    ZFwdArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.in = real_in;
    z.in_zstride = in_zstride;
    z.main = o_main;
    z.nyq = o_nyq;
    z.factor = factor;
    z.lo = lo;
    z.hi = hi;
    int st = dispatch_z_r2c(nz, z, nlines, stream);
    if (st) return st;
    LinePassArgs a{};
    a.fp.type = -1;
    a.n_y = ny;
    a.n_z = nz;
    a.out_scale = 1.0f;
    a.n_geo = 2;
    a.n_grids = 1;
    // pass Y, main block and Nyquist plane
    a.g0 = geo_y_main(nx, ny, H, line_tile_cols(ny), split_xb_log2(nx));
    a.g1 = geo_y_nyq(nx, ny, line_tile_cols(ny));
    a.g1_strided = 1;
    geo_ptrs(a.g0, 0, o_main, o_main);
    geo_ptrs(a.g1, 0, o_nyq, o_nyq);
    if ((st = dispatch_line_pass<-1>(ny, a, 0, stream))) return st;
    // pass X with the normalisation folded into its store
    a.out_scale = out_scale;
    a.g0 = geo_x_main(ny, H, line_tile_cols(nx), split_xb_log2(nx));
    a.g1 = geo_x_nyq(ny, line_tile_cols(nx));
    a.g1_strided = 0;
    geo_ptrs(a.g0, 0, o_main, o_main);
    geo_ptrs(a.g1, 0, o_nyq, o_nyq);
    return dispatch_line_pass<-1>(nx, a, 0, stream);
}

// Pass Z: split_work -> real rows of out_zstride floats.
extern "C" int c21hip_split_z_c2r(const float *split_work, float *real_out, long out_zstride,
                                  int nx, int ny, int nz, void *stream) {
    const long nlines = (long)nx * ny;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out = real_out;
    z.out_zstride = out_zstride;
    z.out_scale = 1.0f;
    return dispatch_z_c2r(nz, z, nlines, (hipStream_t)stream);
}

// Pass Z with the stored value divided by `divisor` (0: no division): the "/ VOLUME" of the
// InitialConditions gathers (InitialConditions.c:687,729,356) folded into the store.
extern "C" int c21hip_split_z_c2r_div(const float *split_work, float *real_out, long out_zstride,
                                      int nx, int ny, int nz, float divisor, void *stream) {
    return c21hip_split_z_c2r_out(split_work, real_out, out_zstride, nx, ny, nz, 1.0f, divisor, 0,
                                  stream);
}

// Pass Z with the general store: v * scale, then / divisor (0: none), then the density floor
// -1 + 1e-7 when floor_density != 0 (the "/ N, clip" of PerturbedField.c:251-276).
extern "C" int c21hip_split_z_c2r_out(const float *split_work, float *real_out, long out_zstride,
                                      int nx, int ny, int nz, float scale, float divisor,
                                      int floor_density, void *stream) {
    const long nlines = (long)nx * ny;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out = real_out;
    z.out_zstride = out_zstride;
    z.out_scale = scale;
    z.out_div = divisor;
    z.out_floor = floor_density;
    return dispatch_z_c2r(nz, z, nlines, (hipStream_t)stream);
}

extern "C" int c21hip_split_xblock_log2(int nx) { return split_xb_log2(nx); }

// Pass Z of the filtered density + its extrema (Eulerian source models with a per-radius
// table).  partials: 2 * nx*ny/16 + 2 * (nx*ny/16384 + 2) doubles; minmax_out[2] on the device.
extern "C" int c21hip_split_z_c2r_minmax(const float *split_work, float *real_out,
                                         long out_zstride, int nx, int ny, int nz,
                                         double *partials, double *minmax_out, void *stream) {
    const long nlines = (long)nx * ny;
    const int nb = (int)(nlines / LZ_PLAIN);
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out = real_out;
    z.out_zstride = out_zstride;
    z.out_scale = 1.0f;
    z.p0 = partials;
    z.p1 = partials + nb;
    int st = dispatch_z_c2r<1>(nz, z, nlines, (hipStream_t)stream);
    if (st) return st;
    double *stage = partials + 2 * (size_t)nb;  // beyond both partial arrays
    if ((st = c21hip_reduce_op(z.p0, nb, 1, stage, minmax_out, stream))) return st;
    return c21hip_reduce_op(z.p1, nb, 2, stage + nb / 1024 + 2, minmax_out + 1, stream);
}

// ... the extrema ALONE (EPI 10: nothing is stored), and the table sweep of the same spectrum with the banded
// barrier (EPI 9) -- round 6, the Eulerian table loop without its delta_R round trip: the c2r runs twice from
// k-space (4 N + 4 N + 1 N bytes read per radius where the store pass, the sweep and the barrier moved
// 4 N + 4 N written + 4 N + 1 N).  Reference: IonisationBox.c:668-699 (extrema), :773-962 (f_coll from the
// table), :1008-1200 (barrier).  Same shapes as the closed form's banded pass Z.
extern "C" int c21hip_z_table_band_supported(int nx, int ny, int nz) {
    return c21hip_z_fcoll_erfc_mask_supported(nx, ny, nz);
}
extern "C" int c21hip_split_z_minmax_only(const float *split_work, int nx, int ny, int nz, double *partials,
                                          double *minmax_out, void *stream) {
    if (!c21hip_z_table_band_supported(nx, ny, nz)) {
        c21hip_set_error("pass Z, extrema only: unsupported box");
        return C21CM_VALUE_ERROR;
    }
    const long nlines = (long)nx * ny;
    const int nb = (int)(nlines / LZ_PLAIN);
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out_zstride = nz;
    z.out_scale = 1.0f;
    z.p0 = partials;
    z.p1 = partials + nb;
    int st;
    {
        KTimeScope kt(13, (hipStream_t)stream);  // kind 13: one-grid pass Z, extrema only
        st = dispatch_z_c2r<10>(nz, z, nlines, (hipStream_t)stream);
    }
    if (st) return st;
    double *stage = partials + 2 * (size_t)nb;  // beyond both partial arrays
    if ((st = c21hip_reduce_op(z.p0, nb, 1, stage, minmax_out, stream))) return st;
    return c21hip_reduce_op(z.p1, nb, 2, stage + nb / 1024 + 2, minmax_out + 1, stream);
}
// nx*ny/16 partial sums of f_coll are left in `partials` for c21hip_eul_band (as the closed form's EPI 7 leaves them)
extern "C" int c21hip_split_z_fcoll_table_band(const float *split_work, float *f_pend, const double *band_dev,
                                               const double *thr_prev_dev, unsigned char *first_cross,
                                               int r_index, int r_prev, int nx, int ny, int nz, int mode,
                                               double tab_min, double tab_width, const float *table_dev,
                                               double *partials, void *stream) {
    if (!c21hip_z_table_band_supported(nx, ny, nz) || r_index <= 0 || r_index >= 255 || r_prev >= 255 ||
        (mode != C21CM_FCOLL_TABLE_LINEAR && mode != C21CM_FCOLL_TABLE_EXP)) {
        c21hip_set_error("pass Z with the table sweep and the banded barrier: unsupported box, mode or radius index");
        return C21CM_VALUE_ERROR;
    }
    const long nlines = (long)nx * ny;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out_zstride = nz;
    z.out_scale = 1.0f;
    z.f_out = f_pend;
    z.band = band_dev;
    z.mean_dev = thr_prev_dev;
    z.mask_rw = first_cross;
    z.r_index = r_index;
    z.r_prev = r_prev;
    z.p0 = partials;
    z.table = table_dev;
    z.tab_min = tab_min;
    z.tab_width = tab_width;
    z.tab_mode = mode;
    KTimeScope kt(14, (hipStream_t)stream);  // kind 14: one-grid pass Z + table f_coll + banded barrier
    return dispatch_z_c2r<9>(nz, z, nlines, (hipStream_t)stream);
}

// Pass Z with the floor-and-scale store of the spin-temperature filter tables
// (SpinTemperatureBox.c:606-629): out = max(v, min_value) * const_factor into dense rows and
// stats_out[3] = {min, max, sum} of the stored values on the device.
// partials: 3 * nx*ny/16 + 2 * (nx*ny/16384 + 2) doubles.
extern "C" int c21hip_split_z_c2r_stats(const float *split_work, float *real_out,
                                        long out_zstride, int nx, int ny, int nz,
                                        double min_value, double const_factor, double *partials,
                                        double *stats_out, void *stream) {
    const long nlines = (long)nx * ny;
    const int nb = (int)(nlines / LZ_PLAIN);
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out = real_out;
    z.out_zstride = out_zstride;
    z.out_scale = 1.0f;
    z.p0 = partials;
    z.p1 = partials + nb;
    z.p2 = partials + 2 * (size_t)nb;
    z.min_value = min_value;
    z.const_factor = const_factor;
    int st = dispatch_z_c2r<3>(nz, z, nlines, (hipStream_t)stream);
    if (st) return st;
    if (!stats_out) return 0;  // deferred: c21hip_batched_stats reduces several windows at once
    double *stage = partials + 3 * (size_t)nb;
    if ((st = c21hip_reduce_op(z.p0, nb, 1, stage, stats_out, stream))) return st;
    if ((st = c21hip_reduce_op(z.p1, nb, 2, stage + nb / 1024 + 2, stats_out + 1, stream)))
        return st;
    return c21hip_reduce_op(z.p2, nb, 0, stage, stats_out + 2, stream);
}

// Pass Z of the filtered density fused with the CONST-ION-EFF closed-form f_coll(delta_R):
// writes the dense f_coll grid and its sum.  partials: >= 2 * nx*ny/16 doubles.
extern "C" int c21hip_split_z_fcoll_erfc(const float *split_work, float *nion_dense, int nx,
                                         int ny, int nz, double growthf, double sigma_min,
                                         double sigma_max, double delta_c, double *partials,
                                         double *sum_out, void *stream) {
    const long nlines = (long)nx * ny;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out_scale = 1.0f;
    z.f_out = nion_dense;
    z.p0 = partials;
    z.delta_c = delta_c;
    z.sig = -1.;
    {
        // hmf.c:1221-1232: float sigmas, float products, double sqrt
        const float ss = (float)sigma_min, sl = (float)sigma_max;
        if (sl > ss) {
            c21hip_set_error("FgtrM requested in a region where M_min > M_max (sigma %g > %g)",
                             (double)sl, (double)ss);
            return C21CM_VALUE_ERROR;
        }
        if (sl != ss)
            z.sig = 1.0 / ((double)(float)growthf * (sqrt(2.) * sqrt((double)(ss * ss - sl * sl))));
    }
    int st;
    {
        KTimeScope kt(12, (hipStream_t)stream);
        st = dispatch_z_c2r<2>(nz, z, nlines, (hipStream_t)stream);
    }
    if (st || !sum_out) return st;  // sum_out NULL: the caller reduces the nlines / 16 partials (c21hip_eul_band)
    return c21hip_reduce_sum(partials, (int)(nlines / LZ_PLAIN), sum_out, stream);
}

// 1: c21hip_split_z_fcoll_erfc_mask serves this box (z-lines on the 16-lane wave kernel)
extern "C" int c21hip_z_fcoll_erfc_mask_supported(int nx, int ny, int nz) {
    return (nz == 512 || nz == 1024) && zw_lines_of(nz, (long)nx * ny) != 0;
}
// c21hip_split_z_fcoll_erfc of THIS radius + the barrier of the previous radius of the loop
// (eulerian_mask_kernel's test on `nion_prev` with its mean *mean_prev_dev, index r_index_prev) in the
// same sweep.  nion_dense and nion_prev are different buffers.
extern "C" int c21hip_split_z_fcoll_erfc_mask(const float *split_work, float *nion_dense,
                                              const float *nion_prev, const double *mean_prev_dev,
                                              unsigned char *first_cross, int r_index_prev, int fix_mean,
                                              double mean_f_coll, int mass_dep_zeta, double f_limit,
                                              double ion_eff, int nx, int ny, int nz, double growthf,
                                              double sigma_min, double sigma_max, double delta_c,
                                              double *partials, double *sum_out, void *stream) {
    if (!c21hip_z_fcoll_erfc_mask_supported(nx, ny, nz) || nion_prev == nion_dense) {
        c21hip_set_error("pass Z with the deferred barrier: unsupported box or aliased f_coll grids");
        return C21CM_VALUE_ERROR;
    }
    const long nlines = (long)nx * ny;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out_scale = 1.0f;
    z.f_out = nion_dense;
    z.f_prev = nion_prev;
    z.mean_dev = mean_prev_dev;
    z.mask_rw = first_cross;
    z.r_index = r_index_prev;
    z.fix_mean = fix_mean;
    z.mean_f_coll = mean_f_coll;
    z.mass_dep_zeta = mass_dep_zeta;
    z.f_limit = f_limit;
    z.ion_eff = ion_eff;
    z.p0 = partials;
    z.delta_c = delta_c;
    z.sig = -1.;
    {
        const float ss = (float)sigma_min, sl = (float)sigma_max;  // hmf.c:1221-1232, as above
        if (sl > ss) {
            c21hip_set_error("FgtrM requested in a region where M_min > M_max (sigma %g > %g)",
                             (double)sl, (double)ss);
            return C21CM_VALUE_ERROR;
        }
        if (sl != ss)
            z.sig = 1.0 / ((double)(float)growthf * (sqrt(2.) * sqrt((double)(ss * ss - sl * sl))));
    }
    int st = dispatch_z_c2r<6>(nz, z, nlines, (hipStream_t)stream);
    if (st) return st;
    return c21hip_reduce_sum(partials, (int)(nlines / LZ_PLAIN), sum_out, stream);
}

// c21hip_split_z_fcoll_erfc of THIS radius with its barrier decided in the same sweep wherever the
// decision does not depend on the exact box mean (EPI 7): band_dev = the two thresholds of this radius
// (eul_band_kernel), f_pend receives the f_coll of the undecided cells (marker 255 in first_cross);
// r_prev >= 0: the radius whose markers are still outstanding, *thr_prev_dev its exact threshold.
extern "C" int c21hip_split_z_fcoll_erfc_band(const float *split_work, float *f_pend,
                                              const double *band_dev, const double *thr_prev_dev,
                                              unsigned char *first_cross, int r_index, int r_prev, int nx,
                                              int ny, int nz, double growthf, double sigma_min,
                                              double sigma_max, double delta_c, double *partials,
                                              double *sum_out, void *stream) {
    if (!c21hip_z_fcoll_erfc_mask_supported(nx, ny, nz) || r_index <= 0 || r_index >= 255 || r_prev >= 255) {
        c21hip_set_error("pass Z with the banded barrier: unsupported box or radius index");
        return C21CM_VALUE_ERROR;
    }
    const long nlines = (long)nx * ny;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(split_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out_scale = 1.0f;
    z.f_out = f_pend;
    z.band = band_dev;
    z.mean_dev = thr_prev_dev;
    z.mask_rw = first_cross;
    z.r_index = r_index;
    z.r_prev = r_prev;
    z.p0 = partials;
    z.delta_c = delta_c;
    z.sig = -1.;
    {
        const float ss = (float)sigma_min, sl = (float)sigma_max;  // hmf.c:1221-1232, as above
        if (sl > ss) {
            c21hip_set_error("FgtrM requested in a region where M_min > M_max (sigma %g > %g)",
                             (double)sl, (double)ss);
            return C21CM_VALUE_ERROR;
        }
        if (sl != ss)
            z.sig = 1.0 / ((double)(float)growthf * (sqrt(2.) * sqrt((double)(ss * ss - sl * sl))));
    }
    int st;
    {
        KTimeScope kt(12, (hipStream_t)stream);  // kind 12: one-grid pass Z + closed-form f_coll (+ banded barrier)
        st = dispatch_z_c2r<7>(nz, z, nlines, (hipStream_t)stream);
    }
    if (st || !sum_out) return st;  // sum_out NULL: the caller reduces the nlines / 16 partials (c21hip_eul_band)
    return c21hip_reduce_sum(partials, (int)(nlines / LZ_PLAIN), sum_out, stream);
}

// Inverse transform of a split spectrum: passes X, Y, Z.
extern "C" int c21hip_split_filter_c2r(const float *split_src, float *split_work, float *real_out,
                                       long out_zstride, int nx, int ny, int nz, double box_len,
                                       double box_len_z, int filter_type, float R, float R_param,
                                       int apply, void *stream) {
    int st = c21hip_split_filter_xy(split_src, split_work, nx, ny, nz, box_len, box_len_z,
                                    filter_type, R, R_param, apply, stream);
    if (st) return st;
    return c21hip_split_z_c2r(split_work, real_out, out_zstride, nx, ny, nz, stream);
}

// Fused pass Z of the density and emissivity grids + f_coll sum + ionisation barrier
// (Lagrangian source grids, radius index > 0).  partials: nx*ny/8 doubles.
extern "C" int c21hip_split_z_ionise_stars(const float *delta_work, const float *stars_work,
                                           unsigned char *first_cross, double *partials,
                                           double *sum_out, int nx, int ny, int nz, int r_index,
                                           double rhocrit_omb, double ion_eff, int mass_dep_zeta,
                                           double f_limit, void *stream) {
    return c21hip_split_z_ionise_stars_xe(delta_work, stars_work, NULL, first_cross, partials,
                                          sum_out, nx, ny, nz, r_index, rhocrit_omb, ion_eff,
                                          mass_dep_zeta, f_limit, stream);
}

// The same with the filtered x_e spectrum of a spin-temperature run as a third grid
// (xe_work == NULL: two grids).  512- and 1024-point z-lines only.
extern "C" int c21hip_split_z_ionise_stars_xe(const float *delta_work, const float *stars_work,
                                              const float *xe_work, unsigned char *first_cross,
                                              double *partials, double *sum_out, int nx, int ny,
                                              int nz, int r_index, double rhocrit_omb,
                                              double ion_eff, int mass_dep_zeta, double f_limit,
                                              void *stream) {
    const long nlines = (long)nx * ny;
    ZFusedArgs a{};
    a.ny = ny;
    a.lb = split_xb_log2(nx);
    a.d_main = reinterpret_cast<const float2 *>(delta_work);
    a.d_nyq = a.d_main + nlines * (nz / 2);
    a.s_main = reinterpret_cast<const float2 *>(stars_work);
    a.s_nyq = a.s_main + nlines * (nz / 2);
    if (xe_work) {
        a.x_main = reinterpret_cast<const float2 *>(xe_work);
        a.x_nyq = a.x_main + nlines * (nz / 2);
    }
    a.first_cross = first_cross;
    a.partials = partials;
    a.rhocrit_omb = rhocrit_omb;
    a.ion_eff = ion_eff;
    a.f_limit = f_limit;
    a.mass_dep_zeta = mass_dep_zeta;
    a.r_index = r_index;
    static const int store_all = [] {
        const char *e = getenv("C21CM_MASK_STORE_ALL");
        return (e && e[0] == '1') ? 1 : 0;
    }();
    a.store_all = store_all;
    static const int reverse = [] {
        const char *e = getenv("C21CM_ZREV");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    a.reverse = reverse;
    int n_partials = 0;
    int st = dispatch_z_fused(nz, a, nlines, (hipStream_t)stream, &n_partials);
    if (st) return st;
    if (!sum_out) return 0;  // deferred: the caller reduces the partials of all radii at once
    return c21hip_reduce_sum(partials, n_partials, sum_out, stream);
}

// The same with a recombination model (CELL_RECOMB): sfr_work = passes X, Y of HaloBox.whalo_sfr;
// nrec: the previous box's cumulative_recombinations (dense; NULL: homogeneous, rec0), g12: dense
// Gamma_12 grid written at first crossings.
extern "C" int c21hip_split_z_ionise_recomb_xe(const float *delta_work, const float *stars_work,
                                               const float *xe_work, const float *nrec, double rec0,
                                               float *g12, unsigned char *first_cross, double *partials,
                                               int nx, int ny, int nz, int r_index, double rhocrit_omb,
                                               double ion_eff, int mass_dep_zeta, double f_limit,
                                               void *stream);
extern "C" int c21hip_split_z_ionise_recomb(const float *delta_work, const float *stars_work,
                                            const float *nrec, double rec0, float *g12,
                                            unsigned char *first_cross,
                                            double *partials, int nx, int ny, int nz, int r_index,
                                            double rhocrit_omb, double ion_eff, int mass_dep_zeta,
                                            double f_limit, void *stream) {
    return c21hip_split_z_ionise_recomb_xe(delta_work, stars_work, nullptr, nrec, rec0, g12, first_cross,
                                           partials, nx, ny, nz, r_index, rhocrit_omb, ion_eff,
                                           mass_dep_zeta, f_limit, stream);
}
// ... with the filtered x_e spectrum of a spin-temperature run as a third line of the barrier kernel
// (xe_work == NULL: none): f zeta > (1 - x_e)(1 + rec), IonisationBox.c:1084-1118
extern "C" int c21hip_split_z_ionise_recomb_xe(const float *delta_work, const float *stars_work,
                                               const float *xe_work, const float *nrec, double rec0,
                                               float *g12, unsigned char *first_cross, double *partials,
                                               int nx, int ny, int nz, int r_index, double rhocrit_omb,
                                               double ion_eff, int mass_dep_zeta, double f_limit,
                                               void *stream) {
    const long nlines = (long)nx * ny;
    ZFusedArgs a{};
    if (xe_work) {
        a.x_main = reinterpret_cast<const float2 *>(xe_work);
        a.x_nyq = a.x_main + nlines * (nz / 2);
    }
    a.ny = ny;
    a.lb = split_xb_log2(nx);
    a.d_main = reinterpret_cast<const float2 *>(delta_work);
    a.d_nyq = a.d_main + nlines * (nz / 2);
    a.s_main = reinterpret_cast<const float2 *>(stars_work);
    a.s_nyq = a.s_main + nlines * (nz / 2);
    a.first_cross = first_cross;
    a.partials = partials;
    a.rhocrit_omb = rhocrit_omb;
    a.ion_eff = ion_eff;
    a.f_limit = f_limit;
    a.mass_dep_zeta = mass_dep_zeta;
    a.r_index = r_index;
#ifdef C21CM_DIAG_BUILD  // (make EXTRA=-DC21CM_DIAG_BUILD: timing diagnostics that change results)
    a.rc = getenv("C21CM_DIAG_RC_NOSTORE") ? 2 : 1;  // (2: no Gamma_12 stores)
#else
    a.rc = 1;
#endif
    a.nrec = nrec;
    a.rec0 = rec0;
    a.g12 = g12;
    a.reverse = 1;
    int n_partials = 0;
    return dispatch_z_fused(nz, a, nlines, (hipStream_t)stream, &n_partials);
}
// ... with the previous snapshot's N_rec FILTERED at this radius (CELL_RECOMB = false) as the third line
// instead of a dense per-cell N_rec: f zeta > 1 + max(N_rec(R), 0) / (1 + delta_R)
extern "C" int c21hip_split_z_ionise_recomb_nrec(const float *delta_work, const float *stars_work,
                                                 const float *nrec_work, float *g12,
                                                 unsigned char *first_cross, double *partials, int nx,
                                                 int ny, int nz, int r_index, double rhocrit_omb,
                                                 double ion_eff, int mass_dep_zeta, double f_limit,
                                                 void *stream) {
    const long nlines = (long)nx * ny;
    ZFusedArgs a{};
    a.x_main = reinterpret_cast<const float2 *>(nrec_work);
    a.x_nyq = a.x_main + nlines * (nz / 2);
    a.x_is_nrec = 1;
    a.ny = ny;
    a.lb = split_xb_log2(nx);
    a.d_main = reinterpret_cast<const float2 *>(delta_work);
    a.d_nyq = a.d_main + nlines * (nz / 2);
    a.s_main = reinterpret_cast<const float2 *>(stars_work);
    a.s_nyq = a.s_main + nlines * (nz / 2);
    a.first_cross = first_cross;
    a.partials = partials;
    a.rhocrit_omb = rhocrit_omb;
    a.ion_eff = ion_eff;
    a.f_limit = f_limit;
    a.mass_dep_zeta = mass_dep_zeta;
    a.r_index = r_index;
    a.rc = 1;
    a.g12 = g12;
    a.reverse = 1;
    int n_partials = 0;
    return dispatch_z_fused(nz, a, nlines, (hipStream_t)stream, &n_partials);
}
// Pass Z of the filtered whalo_sfr of radius r_index: where the first-crossing mask holds r_index,
// g12 (which holds delta_R there, left by c21hip_split_z_ionise_recomb) becomes Gamma_12.
extern "C" int c21hip_split_z_sfr_gamma12(const float *sfr_work, const unsigned char *first_cross,
                                          float *g12, int nx, int ny, int nz, int r_index,
                                          double g12_scale, void *stream) {
    const long nlines = (long)nx * ny;
    const int zwl = zw_lines_of(nz, nlines);
    if (!zwl || zw3_selected(nz, nlines) || nz > 512) return C21CM_VALUE_ERROR;
    const float2 *twH = twiddles(nz / 2);
    const float2 *twN = twiddles(nz);
    if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(sfr_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out = g12;
    z.out_zstride = nz;
    z.out_scale = 1.0f;
    z.const_factor = g12_scale;
    z.mask = first_cross;
    z.r_index = r_index;
    const dim3 grid((unsigned)(nlines / zwl));
    if (nz == 256)
        hipLaunchKernelGGL((zw_c2r_kernel<16, 4, 8>), grid, dim3(kBlock), 0, (hipStream_t)stream, z, twH, twN);
    else
        hipLaunchKernelGGL((zw_c2r_kernel<16, 4, 16>), grid, dim3(kBlock), 0, (hipStream_t)stream, z, twH, twN);
    LAUNCH_CHECK();
    return 0;
}
// Pass Z of the filtered x_e grid of an Eulerian source model with the barrier of the radius fused
// in (EPI 5): f_coll zeta > 1 - x_e(R) into the first-crossing mask; x_e(R) is not written.
extern "C" int c21hip_z_xe_mask_supported(int nx, int ny, int nz) {
    const long nlines = (long)nx * ny;
    return zw_lines_of(nz, nlines) != 0 && !zw3_selected(nz, nlines) && (nz == 256 || nz == 512);
}
extern "C" int c21hip_split_z_xe_mask(const float *xe_work, const float *nion_dense, const double *mean_dev,
                                      unsigned char *first_cross, int nx, int ny, int nz, int r_index,
                                      double mean_f_coll, int fix_mean, int mass_dep_zeta, double f_limit,
                                      double ion_eff, void *stream) {
    if (!c21hip_z_xe_mask_supported(nx, ny, nz)) return C21CM_VALUE_ERROR;
    const long nlines = (long)nx * ny;
    const int zwl = zw_lines_of(nz, nlines);
    const float2 *twH = twiddles(nz / 2);
    const float2 *twN = twiddles(nz);
    if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(xe_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out_zstride = nz;
    z.out_scale = 1.0f;
    z.f_out = const_cast<float *>(nion_dense);
    z.mask_rw = first_cross;
    z.mean_dev = mean_dev;
    z.mean_f_coll = mean_f_coll;
    z.fix_mean = fix_mean;
    z.mass_dep_zeta = mass_dep_zeta;
    z.f_limit = f_limit;
    z.ion_eff = ion_eff;
    z.r_index = r_index;
    KTimeScope kt(11, (hipStream_t)stream);
    const dim3 grid((unsigned)(nlines / zwl));
    if (nz == 256)
        hipLaunchKernelGGL((zw_c2r_kernel<16, 5, 8>), grid, dim3(kBlock), 0, (hipStream_t)stream, z, twH, twN);
    else if (nz == 512)
        hipLaunchKernelGGL((zw_c2r_kernel<16, 5, 16>), grid, dim3(kBlock), 0, (hipStream_t)stream, z, twH, twN);
    else
        return C21CM_VALUE_ERROR;
    LAUNCH_CHECK();
    return 0;
}
// Eulerian table models with an x_e grid, banded barrier (EPI 8): pass Z of the filtered x_e + the table
// sweep of the radius' dense filtered density + the barrier, in one launch.  band_dev = [mf_lo, mf_hi] of
// this radius, mf_prev_dev = the exact mean fix of radius r_prev (whose markers are outstanding; < 0:
// none); f_pend / xe_pend receive (f_coll, clipped x_e) of the undecided cells; nx*ny/16 partial sums of
// f_coll are left in `partials` for c21hip_eul_band.  512-point z-lines.
extern "C" int c21hip_z_xe_fcoll_band_supported(int nx, int ny, int nz) {
    return nz == 512 && zw_lines_of(nz, (long)nx * ny) != 0;
}
extern "C" int c21hip_split_z_xe_fcoll_band(const float *xe_work, const float *delta_fil, long delta_zstride,
                                            float *f_pend, float *xe_pend, const double *band_dev,
                                            const double *mf_prev_dev, unsigned char *first_cross,
                                            int r_index, int r_prev, int mode, double tab_min,
                                            double tab_width, const float *table_dev, int mass_dep_zeta,
                                            double f_limit, double ion_eff, int nx, int ny, int nz,
                                            double *partials, void *stream) {
    if (!c21hip_z_xe_fcoll_band_supported(nx, ny, nz) || r_index <= 0 || r_index >= 255 || r_prev >= 255 ||
        (mode != C21CM_FCOLL_TABLE_LINEAR && mode != C21CM_FCOLL_TABLE_EXP) || delta_zstride % 2) {
        c21hip_set_error("x_e pass Z with the banded barrier: unsupported box, mode or radius index");
        return C21CM_VALUE_ERROR;
    }
    const long nlines = (long)nx * ny;
    const int zwl = zw_lines_of(nz, nlines);
    const float2 *twH = twiddles(nz / 2);
    const float2 *twN = twiddles(nz);
    if (!twH || !twN) return C21CM_MEMORY_ALLOC_ERROR;
    ZPassArgs z{};
    z.ny = ny;
    z.lb = split_xb_log2(nx);
    z.main = reinterpret_cast<const float2 *>(xe_work);
    z.nyq = z.main + nlines * (nz / 2);
    z.out_zstride = delta_zstride;
    z.out_scale = 1.0f;
    z.delta_fil = delta_fil;
    z.table = table_dev;
    z.tab_min = tab_min;
    z.tab_width = tab_width;
    z.tab_mode = mode;
    z.f_out = f_pend;
    z.xe_pend = xe_pend;
    z.band = band_dev;
    z.mean_dev = mf_prev_dev;
    z.mask_rw = first_cross;
    z.mass_dep_zeta = mass_dep_zeta;
    z.f_limit = f_limit;
    z.ion_eff = ion_eff;
    z.r_index = r_index;
    z.r_prev = r_prev;
    z.p0 = partials;
    KTimeScope kt(11, (hipStream_t)stream);
    z.reverse = zw_reverse_default();
    hipLaunchKernelGGL((zw_c2r_kernel<16, 8, 16>), dim3((unsigned)(nlines / zwl)), dim3(kBlock), 0,
                       (hipStream_t)stream, z, twH, twN);
    LAUNCH_CHECK();
    return 0;
}
// ... and with BOTH: the x_e grid of a spin-temperature run and the filtered N_rec of CELL_RECOMB = false
// (four spectra; 512-point z-lines)
extern "C" int c21hip_split_z_ionise_recomb_xe_nrec(const float *delta_work, const float *stars_work,
                                                    const float *xe_work, const float *nrec_work, float *g12,
                                                    unsigned char *first_cross, double *partials, int nx,
                                                    int ny, int nz, int r_index, double rhocrit_omb,
                                                    double ion_eff, int mass_dep_zeta, double f_limit,
                                                    void *stream) {
    if (nz != 512 || !C21X_ZW_TSRC_LEAN) {
        c21hip_set_error("fused pass Z with an x_e grid and a filtered N_rec: 512-point z-lines only");
        return C21CM_VALUE_ERROR;
    }
    const long nlines = (long)nx * ny;
    ZFusedArgs a{};
    a.x_main = reinterpret_cast<const float2 *>(xe_work);
    a.x_nyq = a.x_main + nlines * (nz / 2);
    a.n_main = reinterpret_cast<const float2 *>(nrec_work);
    a.n_nyq = a.n_main + nlines * (nz / 2);
    a.ny = ny;
    a.lb = split_xb_log2(nx);
    a.d_main = reinterpret_cast<const float2 *>(delta_work);
    a.d_nyq = a.d_main + nlines * (nz / 2);
    a.s_main = reinterpret_cast<const float2 *>(stars_work);
    a.s_nyq = a.s_main + nlines * (nz / 2);
    a.first_cross = first_cross;
    a.partials = partials;
    a.rhocrit_omb = rhocrit_omb;
    a.ion_eff = ion_eff;
    a.f_limit = f_limit;
    a.mass_dep_zeta = mass_dep_zeta;
    a.r_index = r_index;
    a.rc = 1;
    a.g12 = g12;
    a.reverse = 1;
    int n_partials = 0;
    return dispatch_z_fused(nz, a, nlines, (hipStream_t)stream, &n_partials);
}
extern "C" int c21hip_z_ionise_recomb_xe_nrec_supported(int nx, int ny, int nz) {
    const long nlines = (long)nx * ny;
    return C21X_ZW_TSRC_LEAN && nz == 512 && zw_lines_of(nz, nlines) != 0 && !zw3_selected(nz, nlines);
}
extern "C" int c21hip_z_ionise_recomb_supported(int nx, int ny, int nz) {
    const long nlines = (long)nx * ny;
    return zw_lines_of(nz, nlines) != 0 && !zw3_selected(nz, nlines) && nz <= 512;
}

// 1: ... also with the x_e grid of a spin-temperature run (16 points per lane: 256- / 512-point z-lines)
extern "C" int c21hip_z_ionise_recomb_xe_supported(int nx, int ny, int nz) {
    return c21hip_z_ionise_recomb_supported(nx, ny, nz) && (nz == 256 || nz == 512);
}

// 1: the fused pass Z can take an x_e grid at this z-line length
extern "C" int c21hip_z_ionise_xe_supported(int nx, int ny, int nz) {
    return zw_lines_of(nz, (long)nx * ny) != 0;
}

// Passes X, Y of ONE grid with window a of the tables built for a two-grid radius
// (c21hip_window_tables / c21hip_split_filter_xy2, buffer `table_slot`): the x_e grid of a
// spin-temperature run shares the density grid's HII_FILTER window.
extern "C" int c21hip_split_filter_xy_shared(const float *src, float *work, int filter_type,
                                             int nx, int ny, int nz, double box_len,
                                             double box_len_z, float R, int apply, int table_slot,
                                             void *stream_) {
    const float *srcs[2] = {src, nullptr};
    float *works[2] = {work, nullptr};
    const int ft[2] = {filter_type, 0};
    const float rp[2] = {0.f, 0.f};
    return filter_xy(srcs, works, 1, nx, ny, nz, box_len, box_len_z, ft, R, rp, apply, stream_, 6,
                     table_slot);
}

// {min, max, sum} of `count` windows in one launch: window i left its per-workgroup partials
// (nb minima, nb maxima, nb sums, in that order) at partials + i * stride; stats_out[3 i ..].
__global__ void __launch_bounds__(kBlock)
batched_stats_kernel(const double *__restrict__ partials, long stride, int nb,
                     double *__restrict__ stats_out) {
    __shared__ double l0[kBlock], l1[kBlock], l2[kBlock];
    const double *p = partials + (long)blockIdx.x * stride;
    double lo = p[0], hi = p[nb], sum = 0.;
    for (int i = threadIdx.x; i < nb; i += kBlock) {
        lo = fmin(lo, p[i]);
        hi = fmax(hi, p[nb + i]);
        sum += p[2 * (long)nb + i];
    }
    l0[threadIdx.x] = lo;
    l1[threadIdx.x] = hi;
    l2[threadIdx.x] = sum;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            l0[threadIdx.x] = fmin(l0[threadIdx.x], l0[threadIdx.x + s]);
            l1[threadIdx.x] = fmax(l1[threadIdx.x], l1[threadIdx.x + s]);
            l2[threadIdx.x] += l2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats_out[3 * blockIdx.x] = l0[0];
        stats_out[3 * blockIdx.x + 1] = l1[0];
        stats_out[3 * blockIdx.x + 2] = l2[0];
    }
}

extern "C" int c21hip_batched_stats(const double *partials, long stride, int nb, int count,
                                    double *stats_out, void *stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(batched_stats_kernel, dim3(count), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, stride, nb, stats_out);
    LAUNCH_CHECK();
    return 0;
}

// workgroup partials the fused pass Z writes for an nx x ny x nz grid
extern "C" int c21hip_z_ionise_partials(int nx, int ny, int nz) {
    const long nlines = (long)nx * ny;
    if (zw3_selected(nz, nlines)) return (int)(nlines / (kBlock / 64));
    if (const int zwl = zw_lines_of(nz, nlines)) return (int)(nlines / (nz == 512 ? C21X_ZW_BLOCK / 16 : zwl));
    return (int)(nlines / LZ_FUSED);
}

// The f_coll sums of `count` radii in ONE launch: block i reduces the n_partials doubles at
// partials + R * stride, R = first - i * step, in a fixed order, then applies the clamp of
// IonisationBox.c:1566-1576 (sums[R], means[R]).  Nothing in the fused Lagrangian loop reads a
// radius' mean, so the three small launches per radius leave the critical path.
__global__ void __launch_bounds__(kBlock)
batched_means_kernel(const double *__restrict__ partials, long stride, int n_partials, int first,
                     int step, double ntot, int mass_dep_zeta, double f_limit,
                     double *__restrict__ sums, double *__restrict__ means) {
    __shared__ double lds[kBlock];
    const int R = first - (int)blockIdx.x * step;
    const double *p = partials + (long)R * stride;
    double acc = 0.;
    for (int i = threadIdx.x; i < n_partials; i += kBlock) acc += p[i];
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[R] = lds[0];
        double m = lds[0] / ntot;
        if (mass_dep_zeta) {
            if (m <= f_limit) m = f_limit;
        } else if (m <= kFractFloatErr) {
            m = kFractFloatErr;
        }
        means[R] = m;
    }
}

extern "C" int c21hip_batched_means(const double *partials, long stride, int n_partials,
                                    int first, int step, int count, double ntot,
                                    int mass_dep_zeta, double f_limit, double *sums,
                                    double *means, void *stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(batched_means_kernel, dim3(count), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, stride, n_partials, first, step, ntot, mass_dep_zeta, f_limit,
                       sums, means);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ single-kernel timing hook
// bench.py's roofline leg: `reps` launches of ONE pass kernel on the caller's stream,
// bracketed by HIP events recorded on that same stream.  Buffers hold pseudo-random data
// (zero-filled operands clock higher and would flatter the number).
__global__ void __launch_bounds__(kBlock)
pattern_fill_kernel(float *__restrict__ buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        buf[i] = ((float)(h & 0xFFFFFF) * (1.0f / 16777216.0f) - 0.5f) * 2e-3f;
    }
}

// kind: 0 pass X and 1 pass Y exactly as the excursion-set loop launches them (two grids,
//       main block + Nyquist plane; pass X streams the window tables of filter_a / filter_b),
//       2 fused pass Z + barrier (two grids), 3 plain pass Z (one grid),
//       4 the window-table kernel of one radius (both windows)
extern "C" int c21hip_bench_pass(int kind, int n, int filter_a, int filter_b, float R,
                                 float R_param_b, double box_len, int reps, void *stream_,
                                 float *ms_out) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!c21hip_native_fft_supported(n, n, n)) return C21CM_VALUE_ERROR;
    const size_t nf = c21hip_split_floats(n, n, n);
    float *a = (float *)c21hip_ws(58, nf * sizeof(float));
    float *b = (float *)c21hip_ws(59, nf * sizeof(float));
    float *real = (float *)c21hip_ws(60, nf * sizeof(float));
    float *real2 = (float *)c21hip_ws(63, nf * sizeof(float));
    unsigned char *mask = (unsigned char *)c21hip_ws(61, (size_t)n * n * n);
    double *partials = (double *)c21hip_ws(62, ((size_t)n * n / 2 + 128) * sizeof(double));
    if (!a || !b || !real || !real2 || !mask || !partials) return C21CM_MEMORY_ALLOC_ERROR;
    hipLaunchKernelGGL(pattern_fill_kernel, dim3(2048), dim3(kBlock), 0, stream, a, nf);
    hipLaunchKernelGGL(pattern_fill_kernel, dim3(2048), dim3(kBlock), 0, stream, b, nf);
    (void)hipMemsetAsync(mask, 0, (size_t)n * n * n, stream);
    const int H = n / 2;
    const long nlines = (long)n * n;
    const float *src[2] = {a, b};
    float *work[2] = {real, real2};
    const int ft[2] = {filter_a, filter_b};
    const float rp[2] = {0.f, R_param_b};
    int st = 0;
    if (kind == 0)  // tables for pass X
        st = filter_xy(src, work, 2, n, n, n, box_len, box_len, ft, R, rp, 1, stream_, 1);
    float *pair2[2] = {nullptr, nullptr};
    if (kind == 6) {  // two-radius pass X: second outputs, tables of R and 0.9 R
        if (n >= 1024) return C21CM_VALUE_ERROR;
        pair2[0] = (float *)c21hip_ws(92, nf * sizeof(float));
        pair2[1] = (float *)c21hip_ws(93, nf * sizeof(float));
        if (!pair2[0] || !pair2[1]) return C21CM_MEMORY_ALLOC_ERROR;
        st = filter_xy_pair(a, real, pair2[0], filter_a, 0.f, b, real2, pair2[1], filter_b,
                            R_param_b, n, n, n, box_len, box_len, R, 0.9f * R, 0, 1, 1, stream_);
    }
    if (kind == 7 || kind == 8) {  // the same passes X with the windows evaluated in the kernel
        if (kind == 8 && n >= 1024) return C21CM_VALUE_ERROR;
        const float radii[2] = {R, 0.9f * R};
        int enabled = 0;
        st = c21hip_wev_prepare(filter_a, 0.f, filter_b, R_param_b, 2, radii, kind == 8 ? 2 : 1, n, n,
                                n, box_len, box_len, kind == 8, &enabled, stream_);
        if (!st && !enabled) st = C21CM_VALUE_ERROR;
        if (kind == 8) {
            pair2[0] = (float *)c21hip_ws(92, nf * sizeof(float));
            pair2[1] = (float *)c21hip_ws(93, nf * sizeof(float));
            if (!pair2[0] || !pair2[1]) return C21CM_MEMORY_ALLOC_ERROR;
        }
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return C21CM_IO_ERROR;
    for (int r = -2; r < reps && !st; r++) {  // two warm-up launches
        if (r == 0) (void)hipEventRecord(e0, stream);
        if (kind == 0 || kind == 7)
            st = filter_xy(src, work, 2, n, n, n, box_len, box_len, ft, R, rp, 1, stream_, 2);
        else if (kind == 8)
            st = filter_xy_pair(a, real, pair2[0], filter_a, 0.f, b, real2, pair2[1], filter_b,
                                R_param_b, n, n, n, box_len, box_len, R, 0.9f * R, 0, 1, 2,
                                stream_);
        else if (kind == 1)
            st = filter_xy(src, work, 2, n, n, n, box_len, box_len, ft, R, rp, 1, stream_, 4);
        else if (kind == 4)
            st = filter_xy(src, work, 2, n, n, n, box_len, box_len, ft, R, rp, 1, stream_, 1);
        else if (kind == 6)
            st = filter_xy_pair(a, real, pair2[0], filter_a, 0.f, b, real2, pair2[1], filter_b,
                                R_param_b, n, n, n, box_len, box_len, R, 0.9f * R, 0, 1, 2,
                                stream_);
        else if (kind == 5)  // pass X without a window (diagnostic)
            st = filter_xy(src, work, 2, n, n, n, box_len, box_len, ft, R, rp, 0, stream_, 2);
        else if (kind == 2)
            st = c21hip_split_z_ionise_stars(a, b, mask, partials, partials + nlines / 4 + 40,
                                             n, n, n, 5, 6.2e9, 1.0, 1, 1e-9, stream);
        else {
            ZPassArgs z{};
            z.ny = n;
            z.lb = split_xb_log2(n);
            z.main = reinterpret_cast<const float2 *>(a);
            z.nyq = z.main + nlines * H;
            z.out = real;
            z.out_zstride = n + 2;
            z.out_scale = 1.0f;
            st = dispatch_z_c2r(n, z, nlines, stream);
        }
    }
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = ms / (float)reps;
    if (kind == 7 || kind == 8) c21hip_wev_release();
    return st;
}

// Generic in-place c2r on the FFTW-style padded layout (used by c21cm_fft_c2r and the
// PerturbedField / InitialConditions drivers for power-of-two grids).
extern "C" int c21hip_native_fft_c2r(float *padded, int nx, int ny, int nz, void *stream) {
    const size_t bytes = c21hip_split_floats(nx, ny, nz) * sizeof(float);
    float *split = (float *)c21hip_ws(48, bytes);
    if (!split) return C21CM_MEMORY_ALLOC_ERROR;
    int st = c21hip_padded_to_split(padded, split, nx, ny, nz, stream);
    if (st) return st;
    return c21hip_split_filter_c2r(split, split, padded, 2 * (long)(nz / 2 + 1), nx, ny, nz, 1.0,
                                   1.0, 0, 0.f, 0.f, 0, stream);
}

