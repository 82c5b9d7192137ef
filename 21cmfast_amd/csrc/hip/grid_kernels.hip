// grid_kernels.hip -- streaming sweeps shared by the three Compute* drivers:
// dense<->padded packing, normalisation, and the fused copy x W(kR) filter sweep.
//
// All of these are HBM-bound (1 read + 1 write per element).  Conventions:
//   * 256-thread workgroups (4 wavefronts of 64), grids capped at 8 blocks per CU
//     x 256 CUs with a grid-stride loop, so the launch fills all 8 XCDs and the
//     tail is short;
//   * padded rows are 2*(nz/2+1) floats long, i.e. only 8-byte aligned, so padded
//     grids are accessed as float2 (one complex k-cell) and dense grids as float2.
#include <hip/hip_runtime.h>

#include <cmath>

#include "c21hip.h"
#include "split_layout.h"
#include "c21cm_abi.h"
#include "ms_window.h"

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

// ---------------------------------------------------------------- pack / unpack
// reference: IonisationBox.c:333-350 -- curr = in*factor (double); out = fmax(fmin(curr,hi),lo)
__global__ void __launch_bounds__(kBlock)
pack_clip_kernel(const float *__restrict__ dense, float *__restrict__ padded, size_t nlines,
                 int nz, int zpad, double factor, double lo, double hi) {
    const size_t total = nlines * (size_t)zpad;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const size_t line = i / (size_t)zpad;
        const int k = (int)(i - line * (size_t)zpad);
        float v = 0.f;
        if (k < nz) {
            double c = (double)dense[line * (size_t)nz + k] * factor;
            v = (float)fmax(fmin(c, hi), lo);
        }
        padded[i] = v;
    }
}

__global__ void __launch_bounds__(kBlock)
unpack_scale_kernel(const float *__restrict__ padded, float *__restrict__ dense, size_t nlines,
                    int nz, int zpad, float scale) {
    const size_t total = nlines * (size_t)nz;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const size_t line = i / (size_t)nz;
        const int k = (int)(i - line * (size_t)nz);
        dense[i] = padded[line * (size_t)zpad + k] * scale;
    }
}

__global__ void __launch_bounds__(kBlock)
divide_kernel(float2 *__restrict__ buf, size_t n2, float divisor) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += (size_t)gridDim.x * kBlock) {
        float2 v = buf[i];
        v.x = __fdiv_rn(v.x, divisor);
        v.y = __fdiv_rn(v.y, divisor);
        buf[i] = v;
    }
}

__global__ void __launch_bounds__(kBlock)
divide_f64_kernel(float2 *__restrict__ buf, size_t n2, double divisor) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += (size_t)gridDim.x * kBlock) {
        float2 v = buf[i];
        v.x = (float)((double)v.x / divisor);
        v.y = (float)((double)v.y / divisor);
        buf[i] = v;
    }
}

__global__ void __launch_bounds__(kBlock)
add_scalar_kernel(float *__restrict__ buf, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock)
        buf[i] = __fadd_rn(buf[i], v);
}

__global__ void __launch_bounds__(kBlock) fill_kernel(float *__restrict__ buf, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock)
        buf[i] = v;
}

__global__ void __launch_bounds__(kBlock)
widen_kernel(const float *__restrict__ in, double *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock)
        out[i] = (double)in[i];
}

// ---------------------------------------------------------------- window functions
// reference: filtering.c:18-32, 80-117.  Evaluated in double like the reference.
__device__ __forceinline__ double w_tophat(double kR) {
    if (kR < 1e-4) return 1 - kR * kR / 10;
    double s, c;
    sincos(kR, &s, &c);
    return 3.0 / (kR * kR * kR) * (s - c * kR);
}
__device__ __forceinline__ double w_sharpk(double kR) { return (kR * 0.413566994 > 1) ? 0. : 1.; }
__device__ __forceinline__ double w_gauss(double kR_sq) { return exp(-0.643 * 0.643 * kR_sq / 2.); }

struct ExpMfpConsts {  // loop invariants of filtering.c:80-104, hoisted to the host
    double R, ratio, ratio2, ratio3, exp_term, ts_0, ts_2;
};
__device__ __forceinline__ double w_exp_mfp(double k, const ExpMfpConsts &c) {
    const double kR = k * c.R;
    if (kR < 1e-4) return c.ts_0 + c.ts_2 * kR * kR;
    double s, co;
    sincos(kR, &s, &co);
    double f = (kR * kR * c.ratio2 + 2 * c.ratio + 1) * c.ratio * co;
    f += (kR * kR * (c.ratio2 - c.ratio3) + c.ratio + 1) * s / kR;
    f *= c.exp_term;
    f -= 2 * c.ratio2;
    const double d = kR * c.ratio * kR * c.ratio + 1;
    f *= -3 * c.ratio / (d * d);
    return f;
}
__device__ __forceinline__ double w_shell(double k, double R_inner, double R_outer) {
    const double kRi = k * R_inner, kRo = k * R_outer;
    if (kRo < 1e-4) {
        const double q = R_inner / R_outer;
        const double q3 = q * q * q;
        return 1. - kRo * kRo / 10 * (q3 * q * q - 1) / (q3 - 1);
    }
    double si, ci, so, co;
    sincos(kRi, &si, &ci);
    sincos(kRo, &so, &co);
    return 3.0 / (kRo * kRo * kRo - kRi * kRi * kRi) * (so - co * kRo - si + ci * kRi);
}

struct FilterParams {
    int nx, ny, nzc;
    int nz0;  // k_z index of column 0 (non-zero for the Nyquist plane of the split layout)
    int lb;   // x-blocked main block of a split spectrum (split_layout.h); 0: lines are x * ny + y
    int type;
    float R, R_param;
    double dkx, dky, dkz;
    ExpMfpConsts mfp;
};

// float-precision wavenumber of grid index n, exactly as filtering.c:335-346
__device__ __forceinline__ float k_of(int n, int dim, double dk) {
    return (n > dim / 2) ? (float)((double)(n - dim) * dk) : (float)((double)n * dk);
}

__device__ __forceinline__ double window_of(const FilterParams &p, float k_mag_sq) {
    // kR is held in float for types 0-2 (filtering.c:331,357-369)
    switch (p.type) {
        case 0: {
            float kR = (float)(sqrt((double)k_mag_sq) * (double)p.R);
            return w_tophat((double)kR);
        }
        case 1: {
            float kR = (float)(sqrt((double)k_mag_sq) * (double)p.R);
            return w_sharpk((double)kR);
        }
        case 2: {
            float kR = __fmul_rn(__fmul_rn(k_mag_sq, p.R), p.R);
            return w_gauss((double)kR);
        }
        case 3:
            return w_exp_mfp(sqrt((double)k_mag_sq), p.mfp);
        default:
            return w_shell(sqrt((double)k_mag_sq), (double)p.R, (double)p.R_param);
    }
}

// One thread per complex k-cell, z fastest -> consecutive lanes touch consecutive
// float2's.  dst = (float)(src * W) per component (complex float times double).
template <bool APPLY>
__global__ void __launch_bounds__(kBlock)
copy_filter_kernel(const float2 *src, float2 *dst, FilterParams p) {  // src may equal dst
    const size_t total = (size_t)p.nx * p.ny * p.nzc;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        float2 v = src[i];
        if (APPLY) {
            const size_t mline = i / (size_t)p.nzc;
            const int n_z = (int)(i - mline * (size_t)p.nzc);
            const size_t line = (size_t)c21_logical_line((long)mline, p.ny, p.lb);
            const int n_x = (int)(line / (size_t)p.ny);
            const int n_y = (int)(line - (size_t)n_x * p.ny);
            const float k_x = k_of(n_x, p.nx, p.dkx);
            const float k_y = k_of(n_y, p.ny, p.dky);
            const float k_z = (float)((double)(n_z + p.nz0) * p.dkz);
            // float adds/muls exactly as written in the reference (no FMA contraction)
            const float k_mag_sq = __fadd_rn(
                __fadd_rn(__fmul_rn(k_x, k_x), __fmul_rn(k_y, k_y)), __fmul_rn(k_z, k_z));
            const double w = window_of(p, k_mag_sq);
            v.x = (float)((double)v.x * w);
            v.y = (float)((double)v.y * w);
        }
        dst[i] = v;
    }
}

// The multiple-scattering window (type 5) has its own kernel: a power series per mode.
__global__ void __launch_bounds__(kBlock)
copy_filter_ms_kernel(const float2 *src, float2 *dst, FilterParams p, MsConsts ms) {
    const size_t total = (size_t)p.nx * p.ny * p.nzc;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        float2 v = src[i];
        const size_t line = i / (size_t)p.nzc;
        const int n_z = (int)(i - line * (size_t)p.nzc);
        const int n_x = (int)(line / (size_t)p.ny);
        const int n_y = (int)(line - (size_t)n_x * p.ny);
        const float k_x = k_of(n_x, p.nx, p.dkx);
        const float k_y = k_of(n_y, p.ny, p.dky);
        const float k_z = (float)((double)(n_z + p.nz0) * p.dkz);
        const float k_mag_sq = __fadd_rn(
            __fadd_rn(__fmul_rn(k_x, k_x), __fmul_rn(k_y, k_y)), __fmul_rn(k_z, k_z));
        const double w = ms_window(sqrt((double)k_mag_sq), ms);
        v.x = (float)((double)v.x * w);
        v.y = (float)((double)v.y * w);
        dst[i] = v;
    }
}

// ---------------------------------------------------------------- floor, scale, statistics
// out = max(in, min_value) * const_factor per cell (float compared with the double floor,
// float x double product rounded to float) + min / max / sum of the stored values: the store
// loop of fill_Rbox_table and one_annular_filter (SpinTemperatureBox.c:606-629,713-731).
// Rows of `in` are in_zstride floats long (padded or dense); out (dense) may be NULL.
__global__ void __launch_bounds__(kBlock)
floor_scale_stats_kernel(const float *__restrict__ in, long in_zstride, float *__restrict__ out,
                         size_t nlines, int nz, double min_value, double const_factor,
                         double *__restrict__ pmin, double *__restrict__ pmax,
                         double *__restrict__ psum) {
    const size_t total = nlines * (size_t)nz;
    double lo = 1e300, hi = -1e300, sum = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const size_t line = i / (size_t)nz;
        const int k = (int)(i - line * (size_t)nz);
        float v = in[line * (size_t)in_zstride + k];
        if ((double)v < min_value) v = (float)min_value;
        v = (float)((double)v * const_factor);
        if (out) out[i] = v;
        lo = fmin(lo, (double)v);
        hi = fmax(hi, (double)v);
        sum += (double)v;
    }
    __shared__ double slo[kBlock / 64], shi[kBlock / 64], ssum[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off, 64));
        hi = fmax(hi, __shfl_down(hi, off, 64));
        sum += __shfl_down(sum, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
        ssum[threadIdx.x >> 6] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) {
            lo = fmin(lo, slo[w]);
            hi = fmax(hi, shi[w]);
            sum += ssum[w];
        }
        pmin[blockIdx.x] = lo;
        pmax[blockIdx.x] = hi;
        psum[blockIdx.x] = sum;
    }
}
}  // namespace

// partials: 3 * C21HIP_PARTIALS doubles; stats_out (device): {min, max, sum}
extern "C" int c21hip_floor_scale_stats(const float *in, long in_zstride, float *out, int nx,
                                        int ny, int nz, double min_value, double const_factor,
                                        double *partials, double *stats_out, void *stream) {
    const size_t nlines = (size_t)nx * ny;
    const int blocks = grid_for(nlines * (size_t)nz);
    double *pmin = partials, *pmax = partials + kMaxBlocks, *psum = partials + 2 * kMaxBlocks;
    hipLaunchKernelGGL(floor_scale_stats_kernel, dim3(blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, in, in_zstride, out, nlines, nz, min_value,
                       const_factor, pmin, pmax, psum);
    LAUNCH_CHECK();
    int st;
    if ((st = c21hip_reduce_op(pmin, blocks, 1, NULL, stats_out, stream))) return st;
    if ((st = c21hip_reduce_op(pmax, blocks, 2, NULL, stats_out + 1, stream))) return st;
    return c21hip_reduce_op(psum, blocks, 0, NULL, stats_out + 2, stream);
}

extern "C" int c21hip_pack_clip(const float *dense, float *padded, int nx, int ny, int nz,
                                double factor, double lo, double hi, void *stream) {
    const int zpad = 2 * (nz / 2 + 1);
    const size_t nlines = (size_t)nx * ny;
    hipLaunchKernelGGL(pack_clip_kernel, dim3(grid_for(nlines * zpad)), dim3(kBlock), 0,
                       (hipStream_t)stream, dense, padded, nlines, nz, zpad, factor, lo, hi);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_unpack_scale(const float *padded, float *dense, int nx, int ny, int nz,
                                   float scale, void *stream) {
    const int zpad = 2 * (nz / 2 + 1);
    const size_t nlines = (size_t)nx * ny;
    hipLaunchKernelGGL(unpack_scale_kernel, dim3(grid_for(nlines * nz)), dim3(kBlock), 0,
                       (hipStream_t)stream, padded, dense, nlines, nz, zpad, scale);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_divide_inplace(float *buf, size_t n, float divisor, void *stream) {
    hipLaunchKernelGGL(divide_kernel, dim3(grid_for(n / 2)), dim3(kBlock), 0, (hipStream_t)stream,
                       (float2 *)buf, n / 2, divisor);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_divide_inplace_f64(float *buf, size_t n, double divisor, void *stream) {
    hipLaunchKernelGGL(divide_f64_kernel, dim3(grid_for(n / 2)), dim3(kBlock), 0,
                       (hipStream_t)stream, (float2 *)buf, n / 2, divisor);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_add_scalar(float *buf, size_t n, float value, void *stream) {
    hipLaunchKernelGGL(add_scalar_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       buf, n, value);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_fill(float *buf, size_t n, float value, void *stream) {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, buf, n,
                       value);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_widen(const float *in, double *out, size_t n, void *stream) {
    hipLaunchKernelGGL(widen_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, in,
                       out, n);
    LAUNCH_CHECK();
    return 0;
}

static int copy_filter_impl(const float *src_c, float *dst_c, int nx, int ny, int nzc, int nz0,
                            double box_len, double box_len_z, int filter_type, float R,
                            float R_param, int apply, void *stream, float R_star = 0.f, int lb = 0);

// filter_box on a spectrum in the split layout (main block [nx][ny][nz/2], x-blocked where
// fft_native.hip blocks it, + Nyquist plane)
extern "C" int c21hip_copy_filter_split(const float *src_split, float *dst_split, int nx, int ny,
                                        int nz, double box_len, double box_len_z, int filter_type,
                                        float R, float R_param, int apply, void *stream) {
    const size_t n_main = 2 * (size_t)nx * ny * (nz / 2);
    if (filter_type == 5 && c21hip_split_xblock_log2(nx)) {
        c21hip_set_error("copy_filter_split: the multiple-scattering window is not built for x-blocked spectra");
        return C21CM_VALUE_ERROR;
    }
    int st = copy_filter_impl(src_split, dst_split, nx, ny, nz / 2, 0, box_len, box_len_z,
                              filter_type, R, R_param, apply, stream, 0.f, c21hip_split_xblock_log2(nx));
    if (st) return st;
    return copy_filter_impl(src_split + n_main, dst_split + n_main, nx, ny, 1, nz / 2, box_len,
                            box_len_z, filter_type, R, R_param, apply, stream);
}

// filter_box with its full argument list (filtering.c:308): R_star matters for type 5 only
extern "C" int c21hip_copy_filter_star(const float *src_c, float *dst_c, int nx, int ny, int nz,
                                       double box_len, double box_len_z, int filter_type, float R,
                                       float R_param, float R_star, int apply, void *stream) {
    return copy_filter_impl(src_c, dst_c, nx, ny, nz / 2 + 1, 0, box_len, box_len_z, filter_type, R,
                            R_param, apply, stream, R_star);
}

extern "C" int c21hip_copy_filter(const float *src_c, float *dst_c, int nx, int ny, int nz,
                                  double box_len, double box_len_z, int filter_type, float R,
                                  float R_param, int apply, void *stream) {
    return copy_filter_impl(src_c, dst_c, nx, ny, nz / 2 + 1, 0, box_len, box_len_z, filter_type, R,
                            R_param, apply, stream);
}

static int copy_filter_impl(const float *src_c, float *dst_c, int nx, int ny, int nzc, int nz0,
                            double box_len, double box_len_z, int filter_type, float R,
                            float R_param, int apply, void *stream, float R_star, int lb) {
    if (apply && (filter_type < 0 || filter_type > 5)) {
        c21hip_set_error("filter type %d is not implemented on the device", filter_type);
        return C21CM_VALUE_ERROR;
    }
    FilterParams p;
    p.lb = lb;
    p.nx = nx;
    p.ny = ny;
    p.nzc = nzc;
    p.nz0 = nz0;
    p.type = filter_type;
    p.R = R;
    p.R_param = R_param;
    p.dkx = 2.0 * M_PI / box_len;
    p.dky = 2.0 * M_PI / box_len;
    p.dkz = 2.0 * M_PI / box_len_z;
    p.mfp = ExpMfpConsts{};
    if (filter_type == 3) {
        // filtering.c:320-322 (float division, double exp) and :83-94
        const double exp_term = exp((double)(-R / R_param));
        const double Rd = (double)R, mfp = (double)R_param;
        const double ratio = mfp / Rd;
        p.mfp.R = Rd;
        p.mfp.ratio = ratio;
        p.mfp.ratio2 = pow(ratio, 2);
        p.mfp.ratio3 = pow(ratio, 3);
        p.mfp.exp_term = exp_term;
        p.mfp.ts_0 = 6 * pow(ratio, 3) -
                     exp_term * (6 * pow(ratio, 3) + 6 * pow(ratio, 2) + 3 * ratio);
        p.mfp.ts_2 = exp_term * (2 * pow(ratio, 2) + 0.5 * ratio) - 2 * p.mfp.ts_0 * pow(ratio, 2);
    }
    const size_t total = (size_t)nx * ny * p.nzc;
    if (apply && filter_type == 5) {
        MsConsts ms;
        ms_fill(ms, R, R_param, R_star);
        hipLaunchKernelGGL(copy_filter_ms_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                           (hipStream_t)stream, (const float2 *)src_c, (float2 *)dst_c, p, ms);
    } else if (apply)
        hipLaunchKernelGGL(copy_filter_kernel<true>, dim3(grid_for(total)), dim3(kBlock), 0,
                           (hipStream_t)stream, (const float2 *)src_c, (float2 *)dst_c, p);
    else
        hipLaunchKernelGGL(copy_filter_kernel<false>, dim3(grid_for(total)), dim3(kBlock), 0,
                           (hipStream_t)stream, (const float2 *)src_c, (float2 *)dst_c, p);
    LAUNCH_CHECK();
    return 0;
}
