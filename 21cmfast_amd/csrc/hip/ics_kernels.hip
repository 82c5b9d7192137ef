// ics_kernels.hip -- ComputeInitialConditions sweeps on MI355X.
//
// reference loops being replaced (src/py21cmfast/src/InitialConditions.c):
//   sample_ic_modes + adj_complex_conj   :26-139   -> sample_modes_kernel
//   compute_f_gradient / _laplacian      :240-297  -> kspace_op_kernel
//   hires_density * V / N packing        :637-653  -> pack_density_kernel
//   2LPT products                        :451-482  -> lpt2_accumulate_kernel
// (the subsampling gathers and the normalisation divide reuse perturb_kernels.hip /
//  grid_kernels.hip.)  All are one-read-one-write HBM sweeps over the DIM^3 grid.
//
// Random modes, two sources for the (a, b) pair of a mode:
//  * a counter-based Philox-4x32-10 keyed by the seed, counter = linear index of the k-cell,
//    Box-Muller in double (C21CM_RNG_PHILOX): independent of launch geometry and N_THREADS;
//  * the reference's own GSL streams drawn on the host in the reference's order
//    (C21CM_RNG_GSL, csrc/host/gsl_stream.c) and read from `deviates[mode]`.
// Either way the Hermitian partners of the k_z = 0 / Nyquist planes take their partner's pair
// (by counter / by index), so the reference's separate symmetrisation pass (adj_complex_conj)
// disappears.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "c21hip.h"
#include "c21cm_abi.h"
#include "split_layout.h"

namespace {
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256 * 8;

inline int grid_for(size_t work_items) {
    size_t b = (work_items + kBlock - 1) / kBlock;
    if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
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

__device__ __forceinline__ void philox4x32_10(uint64_t counter, uint64_t key, uint32_t (&out)[4]) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void gaussian_pair(uint64_t counter, uint64_t seed, double *a,
                                              double *b) {
    uint32_t x[4];
    philox4x32_10(counter, seed, x);
    const double u1 = ((double)(((uint64_t)(x[0] >> 5) << 26) | (x[2] >> 6)) + 0.5) * 0x1p-53;
    const double u2 = ((double)(((uint64_t)(x[1] >> 5) << 26) | (x[3] >> 6)) + 0.5) * 0x1p-53;
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(2.0 * M_PI * u2, &s, &c);
    *a = r * c;
    *b = r * s;
}

// which element (conjugated) a constrained element of the k_z = 0 / Nyquist planes copies
// (InitialConditions.c:58-100); false for a free element
__device__ __forceinline__ bool hermitian_source(int i, int j, int nx, int ny, int *si, int *sj) {
    const int mx = nx / 2, my = ny / 2;
    if (i >= 1 && i < mx) {
        *si = nx - i;
        *sj = (j == 0 || j == my) ? j : ny - j;
        return true;
    }
    if ((i == 0 || i == mx) && j >= 1 && j < my) {
        *si = i;
        *sj = ny - j;
        return true;
    }
    return false;
}

__global__ void __launch_bounds__(kBlock)
sample_modes_kernel(float2 *__restrict__ cbox, int nx, int ny, int nz,
                    const double *__restrict__ pk_by_m, float volume, uint64_t seed,
                    const double2 *__restrict__ deviates, float2 *__restrict__ split_nyq, int lb) {
    const int nzc = nz / 2 + 1, mx = nx / 2, my = ny / 2, mz = nz / 2;
    const size_t total = (size_t)nx * ny * nzc;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const size_t line = t / (size_t)nzc;
        const int n_z = (int)(t - line * (size_t)nzc);
        const int n_x = (int)(line / (size_t)ny);
        const int n_y = (int)(line - (size_t)n_x * ny);
        int gi = n_x, gj = n_y;
        bool conj = false;
        if (n_z == 0 || n_z == mz) {
            int si, sj;
            if (hermitian_source(n_x, n_y, nx, ny, &si, &sj)) {
                gi = si;
                gj = sj;
                conj = true;
            }
        }
        const int ax = gi <= mx ? gi : nx - gi, ay = gj <= my ? gj : ny - gj;
        const long m = (long)ax * ax + (long)ay * ay + (long)n_z * n_z;
        const double p = pk_by_m[m];
        const uint64_t counter = ((uint64_t)gi * ny + gj) * nzc + n_z;
        double a, b;
        if (deviates) {
            // the reference's stream, drawn on the host in its own order (gsl_stream.c); a
            // constrained element takes its partner's pair, which is what adj_complex_conj copies
            const double2 d = deviates[counter];
            a = d.x;
            b = d.y;
        } else {
            gaussian_pair(counter, seed, &a, &b);
        }
        const double amp = sqrt((double)volume * p / 2.0);  // InitialConditions.c:129-130
        float re = (float)(amp * a), im = (float)(amp * b);
        if (conj) im = -im;
        const bool cx = (n_x == 0 || n_x == mx), cy = (n_y == 0 || n_y == my),
                   cz = (n_z == 0 || n_z == mz);
        if (cx && cy && cz) im = 0.f;  // the 7 self-conjugate modes are real (:46-48)
        if (t == 0) re = 0.f;          // zero mode (:50)
        // split_nyq != NULL: cbox is the main block of a split spectrum (round 6: the sampler writes the layout
        // the transforms read -- no padded copy, no conversion sweep)
        if (!split_nyq)
            cbox[t] = make_float2(re, im);
        else if (n_z == mz)
            split_nyq[line] = make_float2(re, im);
        else
            cbox[(size_t)c21_memory_line((long)line, ny, lb) * (size_t)mz + n_z] = make_float2(re, im);
    }
}

// indexing.h:116-120, kept in double here (InitialConditions.c:246-254)
__device__ __forceinline__ double index_to_k(int idx, double len, int dim) {
    const double buf = (idx <= dim / 2) ? (double)idx : (double)(idx - dim);
    return buf * 2. * M_PI / len;
}

// axis1 < 0: out = in * k_axis0 * I / k^2 (gradient); else out = -k_axis0 k_axis1 in / k^2
__global__ void __launch_bounds__(kBlock)
kspace_op_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int nx, int ny, int nz,
                 double len_x, double len_y, double len_z, int axis0, int axis1) {
    const int nzc = nz / 2 + 1;
    const size_t total = (size_t)nx * ny * nzc;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const size_t line = t / (size_t)nzc;
        const int n_z = (int)(t - line * (size_t)nzc);
        const int n_x = (int)(line / (size_t)ny);
        const int n_y = (int)(line - (size_t)n_x * ny);
        const double kvec[3] = {index_to_k(n_x, len_x, nx), index_to_k(n_y, len_y, ny),
                                index_to_k(n_z, len_z, nz)};
        const double k_sq = kvec[0] * kvec[0] + kvec[1] * kvec[1] + kvec[2] * kvec[2];
        const float2 v = in[t];
        float2 o = make_float2(0.f, 0.f);
        if (t != 0) {
            if (axis1 < 0) {
                o.x = (float)(-((double)v.y * kvec[axis0]) / k_sq);
                o.y = (float)(((double)v.x * kvec[axis0]) / k_sq);
            } else {
                const double f = -kvec[axis0] * kvec[axis1];
                o.x = (float)(f * (double)v.x / k_sq);
                o.y = (float)(f * (double)v.y / k_sq);
            }
        }
        out[t] = o;
    }
}


// compute_relative_velocities (InitialConditions.c:141-238), the k-space part of component `axis`:
//   out = in * i * k_axis / |k| * sqrt(P_vcb / P) * c_kms,  DC := 0
// with h(|k|) = sqrt(P_vcb / P) c_kms / |k| tabulated per |k|^2 index m (cubic grids).
__global__ void __launch_bounds__(kBlock)
vcb_op_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int nx, int ny, int nz,
              double len_x, double len_y, double len_z, int axis,
              const double *__restrict__ h_by_m) {
    const int nzc = nz / 2 + 1, mx = nx / 2, my = ny / 2;
    const size_t total = (size_t)nx * ny * nzc;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const size_t line = t / (size_t)nzc;
        const int n_z = (int)(t - line * (size_t)nzc);
        const int n_x = (int)(line / (size_t)ny);
        const int n_y = (int)(line - (size_t)n_x * ny);
        const double kvec[3] = {index_to_k(n_x, len_x, nx), index_to_k(n_y, len_y, ny),
                                index_to_k(n_z, len_z, nz)};
        const int ax = n_x <= mx ? n_x : nx - n_x, ay = n_y <= my ? n_y : ny - n_y;
        const long m = (long)ax * ax + (long)ay * ay + (long)n_z * n_z;
        float2 o = make_float2(0.f, 0.f);
        if (t != 0) {
            const double f = kvec[axis] * h_by_m[m];
            const float2 v = in[t];
            o.x = (float)(-((double)v.y * f));
            o.y = (float)((double)v.x * f);
        }
        out[t] = o;
    }
}

// :203-222 the subsampled square of one component added to lowres_vcb (first = 1: stored);
// last = 1 also finishes with sqrt(.) / VOLUME (:225-232)
__global__ void __launch_bounds__(kBlock)
vcb_accumulate_kernel(const float *__restrict__ src_padded, float *__restrict__ dst, int hx, int hy,
                      int hz, int lx, int ly, int lz, int first, int last, float volume) {
    const size_t total = (size_t)lx * ly * lz, plane = (size_t)ly * lz;
    const double ratio = (double)hx / (double)lx;
    const size_t zpad = 2 * (size_t)(hz / 2 + 1);
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const int i = (int)(t / plane);
        const size_t rem = t - (size_t)i * plane;
        const int j = (int)(rem / (size_t)lz), k = (int)(rem - (size_t)j * lz);
        const int hi = (int)((double)i * ratio + 0.5), hj = (int)((double)j * ratio + 0.5),
                  hk = (int)((double)k * ratio + 0.5);
        const double v = (double)src_padded[(size_t)hk + zpad * ((size_t)hj + (size_t)hy * hi)];
        float acc = first ? (float)(v * v) : (float)((double)dst[t] + v * v);
        if (last) acc = (float)(sqrt((double)acc) / (double)volume);
        dst[t] = acc;
    }
}

// ------------------------------------------------------------------ split-layout IC pipeline
// The k-space side of ComputeInitialConditions on the layout of the native transform
// (fft_native.hip; plain form, nx <= 512): main[(x ny + y) H + k_z] for k_z < H = nz/2 and the
// Nyquist plane nyq[x ny + y].  Working here removes the padded <-> split conversion in front of
// every inverse transform, and lets the low-resolution outputs be computed by FOLDING.

// compute_f_gradient / compute_f_laplacian of one stored mode (InitialConditions.c:240-297)
__device__ __forceinline__ void kop_factor(int n_x, int n_y, int n_z, int nx, int ny, int nz,
                                           double len_x, double len_y, double len_z, int axis0,
                                           int axis1, double *re, double *im) {
    if (axis0 == -1) {  // identity (density)
        *re = 1.;
        *im = 0.;
        return;
    }
    const double kvec[3] = {index_to_k(n_x, len_x, nx), index_to_k(n_y, len_y, ny),
                            index_to_k(n_z, len_z, nz)};
    const double k_sq = kvec[0] * kvec[0] + kvec[1] * kvec[1] + kvec[2] * kvec[2];
    if (n_x == 0 && n_y == 0 && n_z == 0) {
        *re = 0.;
        *im = 0.;
    } else if (axis0 == -2) {  // 1 / k^2: the common factor of every operator below
        *re = 1. / k_sq;
        *im = 0.;
    } else if (axis1 < 0) {  // i k_a / k^2
        *re = 0.;
        *im = kvec[axis0] / k_sq;
    } else {  // -k_a k_b / k^2
        *re = -kvec[axis0] * kvec[axis1] / k_sq;
        *im = 0.;
    }
}

__global__ void __launch_bounds__(kBlock)
split_kop_kernel(const float2 *__restrict__ in_main, const float2 *__restrict__ in_nyq,
                 float2 *__restrict__ out_main, float2 *__restrict__ out_nyq, int nx, int ny,
                 int nz, double len_x, double len_y, double len_z, int axis0, int axis1, int lb) {
    const int H = nz / 2;
    const size_t n_main = (size_t)nx * ny * H, total = n_main + (size_t)nx * ny;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const bool is_main = t < n_main;
        const size_t line = is_main ? t / (size_t)H : t - n_main;  // memory line (main) | logical (Nyquist)
        const int n_z = is_main ? (int)(t - line * (size_t)H) : H;
        const size_t ll = is_main ? (size_t)c21_logical_line((long)line, ny, lb) : line;
        const int n_x = (int)(ll / (size_t)ny), n_y = (int)(ll - (size_t)n_x * ny);
        const float2 v = is_main ? in_main[t] : in_nyq[line];
        double fr, fi;
        kop_factor(n_x, n_y, n_z, nx, ny, nz, len_x, len_y, len_z, axis0, axis1, &fr, &fi);
        const float2 o = make_float2((float)((double)v.x * fr - (double)v.y * fi),
                                     (float)((double)v.x * fi + (double)v.y * fr));
        if (is_main)
            out_main[t] = o;
        else
            out_nyq[line] = o;
    }
}

// Low-resolution output of a high-resolution spectrum WITHOUT the high-resolution transform.
// The reference transforms the (filtered) DIM^3 spectrum back and keeps every f-th cell per axis
// (InitialConditions.c:694-730,318-364).  Decimating the output of an inverse DFT is the
// inverse DFT of the aliased spectrum,
//     x[f m] = sum_{q < n/f} ( sum_{a < f} X[q + a n/f] ) exp(2 pi i q m / (n/f)),
// so the (nx/f)(ny/f)(nz/f) grid follows from ONE read of the stored spectrum (f^3 aliases per
// output mode, the k-space operator applied to each alias at ITS wavenumber) and a transform
// f^3 times smaller.  Aliases with k_z above the stored half come from the Hermitian partner
// conj(X[-k_x, -k_y, nz - k_z]), exactly the element the c2r transform implies there.
__global__ void __launch_bounds__(kBlock)
fold_kernel(const float2 *__restrict__ in_main, const float2 *__restrict__ in_nyq,
            float2 *__restrict__ out_main, float2 *__restrict__ out_nyq, int nx, int ny, int nz,
            int f, double len_x, double len_y, double len_z, int axis0, int axis1, int lb_in,
            int lb_out) {
    const int H = nz / 2, mx = nx / f, my = ny / f, mz = nz / f, Hl = mz / 2;
    const size_t total = (size_t)mx * my * (Hl + 1);
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const size_t line = t / (size_t)(Hl + 1);
        const int qz = (int)(t - line * (size_t)(Hl + 1));
        const int qx = (int)(line / (size_t)my), qy = (int)(line - (size_t)qx * my);
        double acc_r = 0., acc_i = 0.;
        for (int a = 0; a < f; a++)
            for (int b = 0; b < f; b++)
                for (int c = 0; c < f; c++) {
                    int sx = qx + a * mx, sy = qy + b * my, sz = qz + c * mz;
                    const bool partner = sz > H;
                    if (partner) {
                        sx = (nx - sx) % nx;
                        sy = (ny - sy) % ny;
                        sz = nz - sz;
                    }
                    const size_t sl = (size_t)sx * ny + sy;
                    const float2 v = (sz == H) ? in_nyq[sl]
                                               : in_main[(size_t)c21_memory_line((long)sl, ny, lb_in) * (size_t)H + sz];
                    double fr, fi;
                    kop_factor(sx, sy, sz, nx, ny, nz, len_x, len_y, len_z, axis0, axis1, &fr, &fi);
                    const double pr = (double)v.x * fr - (double)v.y * fi;
                    const double pi = (double)v.x * fi + (double)v.y * fr;
                    acc_r += pr;
                    acc_i += partner ? -pi : pi;
                }
        const float2 o = make_float2((float)acc_r, (float)acc_i);
        if (qz == Hl)
            out_nyq[line] = o;
        else
            out_main[(size_t)c21_memory_line((long)line, my, lb_out) * (size_t)Hl + qz] = o;
    }
}

// The same for up to FOUR outputs of one spectrum in one read (round 6): the low-resolution density and the three
// velocity components are folds of the SAME filtered spectrum under the operators 1, i k_x / k^2, i k_y / k^2,
// i k_z / k^2 (InitialConditions.c:299-364,694-730) -- seven fold launches per 2LPT call read the DIM^3 spectrum
// seven times (0.28 ms each at DIM = 512), two of these read it twice.  ops[k]: -1 identity, 0 / 1 / 2 the gradient
// along that axis.  1 / k^2 is formed once per alias (a reciprocal where kop_factor divides per axis: the factor
// moves by at most an ulp of a double before the product is rounded to float).
struct FoldOuts {
    float2 *main[4], *nyq[4];
    int op[4];
    int n;
};
__global__ void __launch_bounds__(kBlock)
fold_multi_kernel(const float2 *__restrict__ in_main, const float2 *__restrict__ in_nyq, FoldOuts o, int nx,
                  int ny, int nz, int f, double len_x, double len_y, double len_z, int lb_in, int lb_out,
                  float tophat_R) {
    const int H = nz / 2, mx = nx / f, my = ny / f, mz = nz / f, Hl = mz / 2;
    const size_t total = (size_t)mx * my * (Hl + 1);
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t line = t / (size_t)(Hl + 1);
        const int qz = (int)(t - line * (size_t)(Hl + 1));
        const int qx = (int)(line / (size_t)my), qy = (int)(line - (size_t)qx * my);
        double acc_r[4] = {0., 0., 0., 0.}, acc_i[4] = {0., 0., 0., 0.};
        for (int a = 0; a < f; a++)
            for (int b = 0; b < f; b++)
                for (int c = 0; c < f; c++) {
                    int sx = qx + a * mx, sy = qy + b * my, sz = qz + c * mz;
                    const bool partner = sz > H;
                    if (partner) {
                        sx = (nx - sx) % nx;
                        sy = (ny - sy) % ny;
                        sz = nz - sz;
                    }
                    const size_t sl = (size_t)sx * ny + sy;
                    float2 v = (sz == H) ? in_nyq[sl]
                                         : in_main[(size_t)c21_memory_line((long)sl, ny, lb_in) * (size_t)H + sz];
                    if (tophat_R > 0.f) {
                        // the real-space top-hat at the low-resolution cell scale on the stored element, as the copy +
                        // filter_box sweep applies it (filtering.c:327-369: float wavenumbers, float squares summed in
                        // float, kR held in float, the window in double, the product rounded to float) -- the
                        // filtered spectrum is then never written (InitialConditions.c:700-703)
                        const float kx = (sx > nx / 2) ? (float)((double)(sx - nx) * (2.0 * M_PI / len_x))
                                                       : (float)((double)sx * (2.0 * M_PI / len_x));
                        const float ky = (sy > ny / 2) ? (float)((double)(sy - ny) * (2.0 * M_PI / len_y))
                                                       : (float)((double)sy * (2.0 * M_PI / len_y));
                        const float kz = (float)((double)sz * (2.0 * M_PI / len_z));
                        const float ksq = __fadd_rn(__fadd_rn(__fmul_rn(kx, kx), __fmul_rn(ky, ky)), __fmul_rn(kz, kz));
                        const float kRf = (float)(sqrt((double)ksq) * (double)tophat_R);
                        const double kR = (double)kRf;
                        double w;
                        if (kR < 1e-4) {
                            w = 1 - kR * kR / 10;
                        } else {
                            double sn, cs;
                            sincos(kR, &sn, &cs);
                            w = 3.0 / (kR * kR * kR) * (sn - cs * kR);
                        }
                        v.x = (float)((double)v.x * w);
                        v.y = (float)((double)v.y * w);
                    }
                    const double kvec[3] = {index_to_k(sx, len_x, nx), index_to_k(sy, len_y, ny),
                                            index_to_k(sz, len_z, nz)};
                    const double k_sq = kvec[0] * kvec[0] + kvec[1] * kvec[1] + kvec[2] * kvec[2];
                    const bool dc = (sx == 0 && sy == 0 && sz == 0);
                    const double inv = dc ? 0. : 1. / k_sq;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (k >= o.n) break;
                        double pr, pi;
                        if (o.op[k] < 0) {
                            pr = (double)v.x;
                            pi = (double)v.y;
                        } else {  // times i g, g = k_a / k^2
                            const double g = kvec[o.op[k]] * inv;
                            pr = -(double)v.y * g;
                            pi = (double)v.x * g;
                        }
                        acc_r[k] += pr;
                        acc_i[k] += partner ? -pi : pi;
                    }
                }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= o.n) break;
            const float2 r = make_float2((float)acc_r[k], (float)acc_i[k]);
            if (qz == Hl)
                o.nyq[k][line] = r;
            else
                o.main[k][(size_t)c21_memory_line((long)line, my, lb_out) * (size_t)Hl + qz] = r;
        }
    }
}

// The 2LPT source from the six second derivatives, all dense (InitialConditions.c:451-493):
// box = 0; for (i,j) in (0,1),(0,2),(1,2): box += phi_ii phi_jj; box -= phi_ij^2, every step
// rounded to float as the reference's in-place loops do; then / norm.
__global__ void __launch_bounds__(kBlock)
lpt2_source_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                   const float *__restrict__ d2, const float *__restrict__ o01,
                   const float *__restrict__ o02, const float *__restrict__ o12,
                   float *__restrict__ out, size_t n4, float norm) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * kBlock) {
        const float4 a0 = reinterpret_cast<const float4 *>(d0)[i];
        const float4 a1 = reinterpret_cast<const float4 *>(d1)[i];
        const float4 a2 = reinterpret_cast<const float4 *>(d2)[i];
        const float4 b01 = reinterpret_cast<const float4 *>(o01)[i];
        const float4 b02 = reinterpret_cast<const float4 *>(o02)[i];
        const float4 b12 = reinterpret_cast<const float4 *>(o12)[i];
        const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w},
                    x2[4] = {a2.x, a2.y, a2.z, a2.w}, y01[4] = {b01.x, b01.y, b01.z, b01.w},
                    y02[4] = {b02.x, b02.y, b02.z, b02.w}, y12[4] = {b12.x, b12.y, b12.z, b12.w};
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float acc = 0.f;
            acc = __fadd_rn(acc, __fmul_rn(x0[e], x1[e]));
            acc = __fsub_rn(acc, __fmul_rn(y01[e], y01[e]));
            acc = __fadd_rn(acc, __fmul_rn(x0[e], x2[e]));
            acc = __fsub_rn(acc, __fmul_rn(y02[e], y02[e]));
            acc = __fadd_rn(acc, __fmul_rn(x1[e], x2[e]));
            acc = __fsub_rn(acc, __fmul_rn(y12[e], y12[e]));
            r[e] = __fdiv_rn(acc, norm);
        }
        reinterpret_cast<float4 *>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// padded = dense * VOLUME / N (float arithmetic, InitialConditions.c:650-651)
__global__ void __launch_bounds__(kBlock)
pack_density_kernel(const float *__restrict__ dense, float *__restrict__ padded, size_t nlines,
                    int nz, int zpad, float volume, float ntot) {
    const size_t total = nlines * (size_t)zpad;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const size_t line = i / (size_t)zpad;
        const int k = (int)(i - line * (size_t)zpad);
        padded[i] =
            (k < nz) ? __fdiv_rn(__fmul_rn(dense[line * (size_t)nz + k], volume), ntot) : 0.f;
    }
}

// box += phi_ii*phi_jj; box -= phi_ij^2, each step rounded to float (InitialConditions.c:470-477)
__global__ void __launch_bounds__(kBlock)
lpt2_accumulate_kernel(float *__restrict__ box, const float *__restrict__ phi_ij_padded,
                       const float *__restrict__ diag_i, const float *__restrict__ diag_j,
                       size_t nlines, int nz, int zpad) {
    const size_t total = nlines * (size_t)nz;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const size_t line = t / (size_t)nz;
        const int k = (int)(t - line * (size_t)nz);
        const size_t f = line * (size_t)zpad + k;
        const double cii = diag_i[t], cjj = diag_j[t], cij = phi_ij_padded[f];
        float b = box[f];
        b = (float)((double)b + cii * cjj);
        b = (float)((double)b - cij * cij);
        box[f] = b;
    }
}
}  // namespace

extern "C" int c21hip_sample_modes(float *cbox, int nx, int ny, int nz, const double *pk_by_m_dev,
                                   float volume, unsigned long long seed,
                                   const double *deviates_dev, void *stream) {
    const size_t total = (size_t)nx * ny * (nz / 2 + 1);
    hipLaunchKernelGGL(sample_modes_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                       (hipStream_t)stream, (float2 *)cbox, nx, ny, nz, pk_by_m_dev, volume,
                       (uint64_t)seed, (const double2 *)deviates_dev, (float2 *)nullptr, 0);
    LAUNCH_CHECK();
    return 0;
}
// the same modes written straight into the split layout (main [lines][nz/2] + Nyquist plane [lines])
extern "C" int c21hip_sample_modes_split(float *split, int nx, int ny, int nz, const double *pk_by_m_dev,
                                         float volume, unsigned long long seed, const double *deviates_dev,
                                         void *stream) {
    const size_t total = (size_t)nx * ny * (nz / 2 + 1);
    float2 *main = (float2 *)split;
    hipLaunchKernelGGL(sample_modes_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, main, nx,
                       ny, nz, pk_by_m_dev, volume, (uint64_t)seed, (const double2 *)deviates_dev,
                       main + (size_t)nx * ny * (nz / 2), c21hip_split_xblock_log2(nx));
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_vcb_op(const float *in_c, float *out_c, int nx, int ny, int nz,
                             double box_len, double box_len_z, int axis, const double *h_by_m_dev,
                             void *stream) {
    const size_t total = (size_t)nx * ny * (nz / 2 + 1);
    hipLaunchKernelGGL(vcb_op_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const float2 *)in_c, (float2 *)out_c, nx, ny, nz, box_len, box_len,
                       box_len_z, axis, h_by_m_dev);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_vcb_accumulate(const float *src_padded, const int hi_dim[3], float *dst,
                                     const int lo_dim[3], int first, int last, float volume,
                                     void *stream) {
    const size_t total = (size_t)lo_dim[0] * lo_dim[1] * lo_dim[2];
    hipLaunchKernelGGL(vcb_accumulate_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                       (hipStream_t)stream, src_padded, dst, hi_dim[0], hi_dim[1], hi_dim[2],
                       lo_dim[0], lo_dim[1], lo_dim[2], first, last, volume);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_kspace_op(const float *in_c, float *out_c, int nx, int ny, int nz,
                                double box_len, double box_len_z, int axis0, int axis1,
                                void *stream) {
    const size_t total = (size_t)nx * ny * (nz / 2 + 1);
    hipLaunchKernelGGL(kspace_op_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const float2 *)in_c, (float2 *)out_c, nx, ny, nz,
                       box_len, box_len, box_len_z, axis0, axis1);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_pack_density(const float *dense, float *padded, int nx, int ny, int nz,
                                   float volume, void *stream) {
    const int zpad = 2 * (nz / 2 + 1);
    const size_t nlines = (size_t)nx * ny;
    hipLaunchKernelGGL(pack_density_kernel, dim3(grid_for(nlines * zpad)), dim3(kBlock), 0,
                       (hipStream_t)stream, dense, padded, nlines, nz, zpad, volume,
                       (float)(nlines * (size_t)nz));
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_lpt2_accumulate(float *box, const float *phi_ij_padded, const float *diag_i,
                                      const float *diag_j, int nx, int ny, int nz, void *stream) {
    const int zpad = 2 * (nz / 2 + 1);
    const size_t nlines = (size_t)nx * ny;
    hipLaunchKernelGGL(lpt2_accumulate_kernel, dim3(grid_for(nlines * nz)), dim3(kBlock), 0,
                       (hipStream_t)stream, box, phi_ij_padded, diag_i, diag_j, nlines, nz, zpad);
    LAUNCH_CHECK();
    return 0;
}

// ---- the reference's random stream: raw words -> deviates (gsl_stream.c) --------------------------
// gsl_ran_ugaussian (polar method): x = 2 u1 - 1, y = 2 u2 - 1, r2 = x^2 + y^2 (accepted on the host),
// value y sqrt(-2 ln r2 / r2).  u = word / 2^32 (mt19937, gfsr4, taus2) or value / 2147483647 (cmrg,
// mrg): the same IEEE double operations as the host's, so x, y, r2 are the host's bits; ln and sqrt
// are the device's (<= 1 ulp from libm's: the deviate can differ in its last bit).
__global__ void __launch_bounds__(kBlock)
gsl_words_kernel(unsigned long long *__restrict__ buf, size_t n, const unsigned char *__restrict__ row_kind,
                 size_t per_row) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const unsigned long long w = buf[i];
        const unsigned kind = row_kind[i / per_row];
        const double a = (double)(unsigned)(w & 0xffffffffu), b = (double)(unsigned)(w >> 32);
        const double u1 = (kind == 2 || kind == 3) ? a / 2147483647.0 : a * (1.0 / 4294967296.0);
        const double u2 = (kind == 2 || kind == 3) ? b / 2147483647.0 : b * (1.0 / 4294967296.0);
        const double x = 2 * u1 - 1, y = 2 * u2 - 1;
        const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
        reinterpret_cast<double *>(buf)[i] = y * sqrt(-2.0 * log(r2) / r2);
    }
}

extern "C" int c21hip_gsl_words_to_deviates(void *buf, size_t n_deviates, const unsigned char *row_kind_dev,
                                            size_t deviates_per_row, void *stream) {
    hipLaunchKernelGGL(gsl_words_kernel, dim3(grid_for(n_deviates)), dim3(kBlock), 0, (hipStream_t)stream,
                       (unsigned long long *)buf, n_deviates, row_kind_dev, deviates_per_row);
    LAUNCH_CHECK();
    return 0;
}

// ---- split-layout pipeline entry points (plain layout, or x-blocked where fft_native.hip blocks: nx >= 1024)
extern "C" int c21hip_split_kop(const float *in_split, float *out_split, int nx, int ny, int nz,
                                double box_len, double box_len_z, int axis0, int axis1,
                                void *stream) {
    const size_t n_main = (size_t)nx * ny * (nz / 2);
    const float2 *im = reinterpret_cast<const float2 *>(in_split);
    float2 *om = reinterpret_cast<float2 *>(out_split);
    hipLaunchKernelGGL(split_kop_kernel, dim3(grid_for(n_main + (size_t)nx * ny)), dim3(kBlock), 0,
                       (hipStream_t)stream, im, im + n_main, om, om + n_main, nx, ny, nz, box_len,
                       box_len, box_len_z, axis0, axis1, c21hip_split_xblock_log2(nx));
    LAUNCH_CHECK();
    return 0;
}

// hi (nx, ny, nz) split spectrum -> lo (nx/f, ny/f, nz/f) split spectrum of the decimated field,
// the operator (axis0, axis1) applied per alias; axis0 < 0: none
extern "C" int c21hip_split_fold(const float *hi_split, float *lo_split, int nx, int ny, int nz,
                                 int f, double box_len, double box_len_z, int axis0, int axis1,
                                 void *stream) {
    if (f < 1 || nx % f || ny % f || nz % f || (nz / f) % 2) {
        c21hip_set_error("fold: %dx%dx%d is not divisible by %d", nx, ny, nz, f);
        return C21CM_VALUE_ERROR;
    }
    const size_t n_main = (size_t)nx * ny * (nz / 2);
    const int mx = nx / f, my = ny / f, mz = nz / f;
    const size_t l_main = (size_t)mx * my * (mz / 2);
    const float2 *im = reinterpret_cast<const float2 *>(hi_split);
    float2 *om = reinterpret_cast<float2 *>(lo_split);
    hipLaunchKernelGGL(fold_kernel, dim3(grid_for((size_t)mx * my * (mz / 2 + 1))), dim3(kBlock), 0,
                       (hipStream_t)stream, im, im + n_main, om, om + l_main, nx, ny, nz, f,
                       box_len, box_len, box_len_z, axis0, axis1, c21hip_split_xblock_log2(nx),
                       c21hip_split_xblock_log2(mx));
    LAUNCH_CHECK();
    return 0;
}

// hi spectrum -> n_out (<= 4) folded lo spectra, ops[k] = -1 (identity) or the gradient axis 0 / 1 / 2
// tophat_R > 0: the top-hat window of that radius is applied to every stored element first (fold(filter(X)))
extern "C" int c21hip_split_fold_multi(const float *hi_split, float *const lo_split[4], const int ops[4], int n_out,
                                       int nx, int ny, int nz, int f, double box_len, double box_len_z,
                                       float tophat_R, void *stream) {
    if (f < 1 || nx % f || ny % f || nz % f || (nz / f) % 2 || n_out < 1 || n_out > 4) {
        c21hip_set_error("fold: %dx%dx%d / %d with %d outputs is not supported", nx, ny, nz, f, n_out);
        return C21CM_VALUE_ERROR;
    }
    const size_t n_main = (size_t)nx * ny * (nz / 2);
    const int mx = nx / f, my = ny / f, mz = nz / f;
    const size_t l_main = (size_t)mx * my * (mz / 2);
    const float2 *im = reinterpret_cast<const float2 *>(hi_split);
    FoldOuts o{};
    o.n = n_out;
    for (int k = 0; k < n_out; k++) {
        if (ops[k] < -1 || ops[k] > 2 || !lo_split[k]) {
            c21hip_set_error("fold: operator %d / NULL output", ops[k]);
            return C21CM_VALUE_ERROR;
        }
        o.main[k] = reinterpret_cast<float2 *>(lo_split[k]);
        o.nyq[k] = o.main[k] + l_main;
        o.op[k] = ops[k];
    }
    hipLaunchKernelGGL(fold_multi_kernel, dim3(grid_for((size_t)mx * my * (mz / 2 + 1))), dim3(kBlock), 0,
                       (hipStream_t)stream, im, im + n_main, o, nx, ny, nz, f, box_len, box_len, box_len_z,
                       c21hip_split_xblock_log2(nx), c21hip_split_xblock_log2(mx), tophat_R);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_lpt2_source(const float *const diag[3], const float *const off[3],
                                  float *out, size_t n, float norm, void *stream) {
    if (n % 4) {
        c21hip_set_error("lpt2 source: cell count must be a multiple of 4");
        return C21CM_VALUE_ERROR;
    }
    hipLaunchKernelGGL(lpt2_source_kernel, dim3(grid_for(n / 4)), dim3(kBlock), 0,
                       (hipStream_t)stream, diag[0], diag[1], diag[2], off[0], off[1], off[2], out,
                       n / 4, norm);
    LAUNCH_CHECK();
    return 0;
}
