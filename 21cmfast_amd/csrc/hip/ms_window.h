// ms_window.h -- the multiple-scattering window of the Lyman-alpha shells (filter type 5).
//
// reference: src/py21cmfast/src/filtering.c:119-306 (arXiv:2601.14360, Eqs. 11, 25-32, E7-E8):
//   W(k) = [R_o^3 F(k R_o; a_o, b_o) - R_i^3 F(k R_i; a_i, b_i)] / (R_o^3 - R_i^3),
//   F = 2F3((a+2)/2, (a+3)/2; 5/2, (a+b+2)/2, (a+b+3)/2; -(kR)^2/4).
// F is a power series below kR = 30 and an asymptotic form above; everything in the asymptotic
// form that does not depend on k (five Gamma functions, two reciprocal Gammas, the phase) is
// folded into MsSide on the host once per shell, so the device evaluates two pow, one sincos
// and a handful of multiplies per mode.  The same inline code serves the host-side exports
// (hyper_2F3, compute_mu/eta_for_multiple_scattering) and the window kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

struct MsSide {
    double alpha, beta;
    double phase;     // pi (2 + beta) / 2
    double sin_coef;  // 1 + (alpha - 1) beta
    double scale;     // Gamma(5/2)/sqrt(pi) * Gamma(b2)/Gamma(a1) * Gamma(b3)/Gamma(a2)
    double decay1, decay2;  // coefficients of (kR/2)^-(alpha+2) and (kR/2)^-(alpha+3)
};

struct MsConsts {
    MsSide inner, outer;
    double Ri, Ro;    // radii as the doubles filter_box's float arguments promote to
    double Ri3, Ro3;
};

// ---- Eqs. (29), (30): fits of mu and eta as functions of x_em = R / R_star
__host__ __device__ inline double ms_poly5(double z, double c5, double c4, double c3, double c2,
                                           double c1, double c0) {
    return c5 * pow(z, 5) + c4 * pow(z, 4) + c3 * pow(z, 3) + c2 * z * z + c1 * z + c0;
}
__host__ __device__ inline double ms_mu(double x_em) {
    const double z = log10(x_em);
    if (x_em > 30) return 1. - 1.0478 * pow(x_em, -0.7266);
    if (x_em > 3.) return ms_poly5(z, -0.104, 0.4867, -0.8217, 0.4889, 0.264, 0.518);
    if (x_em > 0.2) return ms_poly5(z, -0.0285, 0.087, -0.1205, -0.0456, 0.3787, 0.5285);
    return 0.3982 * pow(x_em, 0.1592);
}
__host__ __device__ inline double ms_eta(double x_em) {
    const double z = log10(x_em);
    if (x_em > 20.) return 1. - 2.804 * pow(x_em, -1.242);
    if (x_em > 3.) return ms_poly5(z, 2.17, -8.832, 13.579, -10.04, 4.166, -0.17);
    if (x_em > 0.2) return ms_poly5(z, 0.352, -0.0516, -0.293, 0.342, 0.582, 0.266);
    return 0.4453 * pow(x_em, 1.296);
}

// 1 / Gamma(x) with its zeros at the poles of Gamma
inline double ms_rgamma(double x) {
    if (x <= 0. && x == floor(x)) return 0.;
    return 1. / tgamma(x);
}

// the k-independent part of the large-argument expansion (filtering.c:189-254)
inline void ms_side_fill(MsSide &s, double alpha, double beta) {
    s.alpha = alpha;
    s.beta = beta;
    s.phase = M_PI * (2. + beta) / 2.;
    s.sin_coef = 1. + (alpha - 1.) * beta;
    const double a1 = (2. + alpha) / 2., a2 = (3. + alpha) / 2., b1 = 5. / 2.;
    const double b2 = (2. + alpha + beta) / 2., b3 = (3. + alpha + beta) / 2.;
    const double g_a1 = tgamma(a1), g_a2 = tgamma(a2);
    double r21, r32;
    if (a1 < 20.) {
        r21 = tgamma(b2) / g_a1;
        r32 = tgamma(b3) / g_a2;
    } else {  // Stirling ratio, the Gammas themselves would overflow
        const double y = beta / 2;
        r21 = pow(a1, y) * exp((a1 + y - 0.5) * (y / a1 - y * y / (2. * a1 * a1) +
                                                 y * y * y / (3. * a1 * a1 * a1)) -
                               y);
        r32 = pow(a2, y) * exp((a2 + y - 0.5) * (y / a2 - y * y / (2. * a2 * a1) +
                                                 y * y * y / (3. * a2 * a2 * a2)) -
                               y);
    }
    s.scale = 3. / 4. * r21 * r32;
    if (alpha < 10.) {
        s.decay1 = M_PI * g_a1 * ms_rgamma(b1 - a1) / tgamma(b2 - a1) / tgamma(b3 - a1);
        s.decay2 = -2. * M_PI * g_a2 * ms_rgamma(b1 - a2) * ms_rgamma(b2 - a2) / tgamma(b3 - a2);
    } else {
        s.decay1 = 0.;
        s.decay2 = 0.;
    }
}

// Eqs. (25), (28): alpha, beta of both shell edges (filtering.c:162-187)
inline void ms_fill(MsConsts &c, float R_inner, float R_outer, float R_star) {
    double ai = 1., bi = 1., ao = 1., bo = 0.;  // R_star == 0: no scattering
    if (R_star != 0.f) {
        const double xi = (double)R_inner / (double)R_star, xo = (double)R_outer / (double)R_star;
        const double mu_i = ms_mu(xi), eta_i = ms_eta(xi), mu_o = ms_mu(xo), eta_o = ms_eta(xo);
        ai = (1. / eta_i - 1.) / pow(1. / mu_i - 1., 2);
        bi = (1. / eta_i - 1.) / (1. / mu_i - 1.);
        ao = (1. / eta_o - 1.) / pow(1. / mu_o - 1., 2);
        bo = (1. / eta_o - 1.) / (1. / mu_o - 1.);
    }
    ms_side_fill(c.inner, ai, bi);
    ms_side_fill(c.outer, ao, bo);
    c.Ri = (double)R_inner;
    c.Ro = (double)R_outer;
    c.Ri3 = pow(c.Ri, 3.);
    c.Ro3 = pow(c.Ro, 3.);
}

// F(kR) of one shell edge (filtering.c:258-293)
__host__ __device__ inline double ms_hyper(double kR, const MsSide &s) {
    if (s.beta == 0. || kR >= 30.) {
        const double F_sl = 3.0 / (kR * kR * kR) * (sin(kR) - cos(kR) * kR);
        if (s.beta == 0.) return F_sl;  // no neutral hydrogen: straight lines
        const double half = kR / 2.;
        const double arg = kR - s.phase;
        double F = (cos(arg) - s.sin_coef / kR * sin(arg)) / pow(half, s.beta + 2.);
        F += s.decay1 / pow(half, s.alpha + 2.) + s.decay2 / pow(half, s.alpha + 3.);
        F *= s.scale;
        return (fabs(F) < fabs(F_sl)) ? F : F_sl;
    }
    double sum = 0., term = 1.;
    for (int n = 1; n < 1000; n++) {
        sum += term;
        term *= -1. / (1. + s.beta / (s.alpha + 2. * n)) / (1. + s.beta / (s.alpha + 1 + 2. * n)) *
                kR * kR / (2. * n) / (2. * n + 3.);
        if (fabs(term) < fabs(sum) * 1e-4) break;
    }
    return sum;
}

__host__ __device__ inline double ms_window(double k, const MsConsts &c) {
    double W = c.Ro3 * ms_hyper(k * c.Ro, c.outer) - c.Ri3 * ms_hyper(k * c.Ri, c.inner);
    W /= c.Ro3 - c.Ri3;
    return W;
}
