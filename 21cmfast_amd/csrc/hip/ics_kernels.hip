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
// Random modes: a counter-based Philox-4x32-10 keyed by the seed, counter = linear index
// of the k-cell, Box-Muller in double.  Being counter-based, the Hermitian partners of
// the k_z = 0 / Nyquist planes are generated directly from their partner's counter, so
// the reference's separate symmetrisation pass (adj_complex_conj) disappears, and the
// realisation does not depend on launch geometry (the reference's depends on N_THREADS).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "c21hip.h"
#include "c21cm_abi.h"

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
                    const double *__restrict__ pk_by_m, float volume, uint64_t seed) {
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
        gaussian_pair(counter, seed, &a, &b);
        const double amp = sqrt((double)volume * p / 2.0);  // InitialConditions.c:129-130
        float re = (float)(amp * a), im = (float)(amp * b);
        if (conj) im = -im;
        const bool cx = (n_x == 0 || n_x == mx), cy = (n_y == 0 || n_y == my),
                   cz = (n_z == 0 || n_z == mz);
        if (cx && cy && cz) im = 0.f;  // the 7 self-conjugate modes are real (:46-48)
        if (t == 0) re = 0.f;          // zero mode (:50)
        cbox[t] = make_float2(re, im);
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
                                   float volume, unsigned long long seed, void *stream) {
    const size_t total = (size_t)nx * ny * (nz / 2 + 1);
    hipLaunchKernelGGL(sample_modes_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                       (hipStream_t)stream, (float2 *)cbox, nx, ny, nz, pk_by_m_dev, volume,
                       (uint64_t)seed);
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
