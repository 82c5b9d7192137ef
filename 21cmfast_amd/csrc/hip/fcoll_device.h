// fcoll_device.h -- per-cell collapsed-fraction helpers of the Eulerian source models, shared by
// ionize_kernels.hip (stand-alone sweep) and fft_native.hip (fused into pass Z).
#ifndef C21HIP_FCOLL_DEVICE_H
#define C21HIP_FCOLL_DEVICE_H

#include <hip/hip_runtime.h>

namespace {
constexpr double kFractFloatErr = 1e-7;  // reference: Constants.h FRACT_FLOAT_ERR

// reference: hmf.c:1187-1203 (float in, double polynomial, float out).  The result is a float
// and the formula itself is a 1.2e-7 approximation of erfc, so the two expensive pieces are
// evaluated to float accuracy instead of with ~55 fp64 instructions: 1/(1 + q/2) from the
// hardware reciprocal plus one Newton step (1e-14), and exp(u) as 2^n * exp2(frac) with the
// hardware exp2 on the fraction (2e-7 relative).  The polynomial stays in double.  The pass-Z
// sweep that carries this was fp64-ALU bound (0.48 ms against 0.20 ms for the plain store).
// Round 5 (C21X_ERFC_F32, default): the whole evaluation in fp32 with explicit FMAs -- Horner in float, the
// reciprocal with one float Newton step, exp(u) = 2^n exp2(fma(u, log2 e, -n)).  The closed-form pass Z was
// fp64-issue bound (1600 of its 5300 instructions were the double polynomial, 4.4 cycles each against 2.7):
// against the double polynomial the float result moves by <= 7e-7 relative for x < 3, i.e. where f_coll zeta
// can reach a barrier (2e-6 at x = 6, f ~ 1e-17), mean 1e-7 -- the size of the fit's own error.  Every
// sweep of the closed form (dense, banded, deferred, stand-alone) calls this one function.
#ifndef C21X_ERFC_F32
#define C21X_ERFC_F32 1
#endif
#if C21X_ERFC_F32
__device__ __forceinline__ float erfcc_f(float x) {
    const float q = fabsf(x);
    const float d = __fmaf_rn(0.5f, q, 1.0f);
    const float t0 = __builtin_amdgcn_rcpf(d);
    const float t = __fmaf_rn(t0, __fmaf_rn(-d, t0, 1.0f), t0);
    float p = 0.17087277f;
    p = __fmaf_rn(p, t, -0.82215223f);
    p = __fmaf_rn(p, t, 1.4885159f);
    p = __fmaf_rn(p, t, -1.13520398f);
    p = __fmaf_rn(p, t, 0.2788681f);
    p = __fmaf_rn(p, t, -0.1862881f);
    p = __fmaf_rn(p, t, 0.0967842f);
    p = __fmaf_rn(p, t, 0.374092f);
    p = __fmaf_rn(p, t, 1.0000237f);
    const float u = __fmaf_rn(-q, q, __fmaf_rn(t, p, -1.2655122f));
    const float n = floorf(u * 1.44269504f);
    // the fraction from u itself (one rounding), the low part of log2 e added back
    const float fr = __fmaf_rn(u, 1.925963033500011e-8f, __fmaf_rn(u, 1.4426950216293335f, -n));
    const float e = (n < -160.f) ? 0.f : ldexpf(__builtin_amdgcn_exp2f(fr), (int)n);
    const float ans = t * e;
    return x >= 0.0f ? ans : 2.0f - ans;
}
#else
__device__ __forceinline__ float erfcc_f(float x) {
    const double q = fabs((double)x);
    const double d = 1.0 + 0.5 * q;
    const double t0 = (double)__builtin_amdgcn_rcpf((float)d);
    const double t = t0 * (2.0 - d * t0);
    const double u = -q * q - 1.2655122 +
                     t * (1.0000237 +
                          t * (0.374092 +
                               t * (0.0967842 +
                                    t * (-0.1862881 +
                                         t * (0.2788681 +
                                              t * (-1.13520398 +
                                                   t * (1.4885159 +
                                                        t * (-0.82215223 + t * 0.17087277))))))));
    const double w = u * 1.4426950408889634;  // log2(e)
    const double n = floor(w);
    // below 2^-160 the float result is zero anyway; clamping keeps the int conversion defined
    const float e = (n < -160.) ? 0.f
                                : ldexpf(__builtin_amdgcn_exp2f((float)(w - n)), (int)n);
    const float ans = (float)(t * (double)e);
    return x >= 0.0f ? ans : 2.0f - ans;
}
#endif

// reference: hmf.c:1205-1241.  sig (from the float sigmas) is precomputed on the host.
__device__ __forceinline__ double fgtrm_bias_fast(float growthf, float del_bias, double sig,
                                                  double delta_c) {
    const double del = (delta_c - (double)del_bias) / (double)growthf;
    const double x = del / (sqrt(2.) * sig);
    if (x < 0) return 1.0;
    return (double)erfcc_f((float)x);
}

// The same with the two cell-independent divisions turned into multiplications by
// inv = 1 / (growthf * sqrt(2) * sig), computed once on the host in double: the argument x
// agrees with the two-division form to 2 ulp (double) before it is rounded to float, so the
// float handed to erfcc differs for about one cell in 1e8.  The sweep is fp64-ALU bound and
// a division costs as much as the whole polynomial.
__device__ __forceinline__ double fgtrm_bias_fast_inv(float del_bias, double inv, double delta_c) {
    const double x = (delta_c - (double)del_bias) * inv;
    if (x < 0) return 1.0;
    return (double)erfcc_f((float)x);
}

// exp(u) to float accuracy (2e-7 relative): 2^n * exp2(frac) with the hardware exp2; results
// that are stored as float do not need the ~40 fp64 instructions of the library routine.
__device__ __forceinline__ double exp_f32acc(double u) {
    const double w = u * 1.4426950408889634;  // log2(e)
    const double n = floor(w);
    if (!(n > -160.) || !(n < 1000.)) return exp(u);  // underflow / overflow / NaN: exact path
    return (double)ldexpf(__builtin_amdgcn_exp2f((float)(w - n)), (int)n);
}

// reference: interpolation.c:123-131
__device__ __forceinline__ double eval_table_f(double x, double x_min, double x_width,
                                               const float *y_arr) {
    const int idx = (int)floor((x - x_min) / x_width);
    const double table_val = x_min + x_width * (double)(float)idx;
    const double interp_point = (x - table_val) / x_width;
    return (double)y_arr[idx] * (1 - interp_point) + (double)y_arr[idx + 1] * interp_point;
}
// The same with the two cell-independent divisions as multiplications by inv_width = 1 / x_width
// (round 5; the f_coll sweeps of the table modes were bound by them: two IEEE double divisions are ~35 of
// the ~100 fp64 instructions a cell cost, 336 us per 512^3 sweep where its bytes need 130).  The quotients
// agree with the divisions to 2 ulp of a double: the interpolation weight moves by 3e-16, i.e. the float
// the result is stored / compared as differs for about one cell in 1e8 -- and an index that lands on the
// other side of a node (x exactly on it) interpolates to the same value from the neighbouring interval.
// All three table sweeps (dense, banded, pass Z EPI 8) use this form, so they still agree bit for bit.
// ... and with the interpolation weight taken as the fraction of that quotient (q - idx instead of
// (x - (x_min + w idx)) / w: the same number up to 1e-16 idx) and the lerp as one FMA: 9 fp64 instructions
// instead of 16 (the sweeps are fp64-issue bound: 4.4 cycles per instruction and wave, section 4).
__device__ __forceinline__ double eval_table_f_inv(double x, double x_min, double x_width, double inv_width,
                                                   const float *y_arr) {
    (void)x_width;
    const double q = (x - x_min) * inv_width;  // >= 0: the table starts below the box minimum
    const int idx = (int)q;
    const double interp_point = q - (double)idx;
    const double y0 = (double)y_arr[idx];
    return fma(interp_point, (double)y_arr[idx + 1] - y0, y0);
}

// both clips applied to the filtered density, IonisationBox.c:689 then :803
__device__ __forceinline__ float clip_delta_eulerian(float v) {
    v = fmaxf(fminf(v, 1e6f), -1.f);  // (1e6 is a float: the same value as (float)fmin((double)v, 1e6))
    return fmaxf(v, (float)(-1. + kFractFloatErr));
}
__device__ __forceinline__ float clip_delta(float v) {
    return fmaxf(v, (float)(-1. + kFractFloatErr));
}
}  // namespace
#endif
