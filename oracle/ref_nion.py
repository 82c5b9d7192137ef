"""TEST INFRASTRUCTURE ONLY (never imported by the product): numpy / scipy restatement of the per-cell
conditional N_ion integral of SOURCE_MODEL = E-INTEGRAL WITHOUT interpolation tables.

What it restates (all paths relative to /root/reference/src/py21cmfast/src/):
* IonisationBox.c:866-880 -> interp_tables.c:986-1001 (EvaluateNion_Conditional, no-table branch: the
  turnover mass is sc->mturn_a_nofb without mini-halos) -> hmf.c:1106-1140 (Nion_ConditionalM: zero for
  lnM1 >= lnM_cond, one halo of the condition mass above MAX_DELTAC_FRAC of the barrier, the
  Gauss-Legendre integral otherwise);
* hmf.c:659-694 (gauleg: Newton iteration on the Legendre polynomial from the cos() guess, EPS2 = 3e-11,
  NGL_INT = 100 nodes in ln M) and :710-730 (IntegratedNdM_GL: sum of w_i f(x_i));
* hmf.c:541-543, 462-468 (c_nion_integrand = nion_fraction x conditional mass function),
  scaling_relations.c:225-231 (log_scaling_PL_limit), hmf.c:234-287 (st_taylor_factor,
  dNdM_conditional_ST), hmf.c:151-172 (sheth_delc_fixed, get_delta_crit for HMF = ST).

sigma(M) and d sigma^2 / dM come from oracle/ref_scalars.py's own Eisenstein-Hu power spectrum integrated
by scipy -- nothing here calls the library under test.  The node data (x_i, w_i, sigma_i, dsigma^2/dM_i,
nion_fraction_i) do not depend on the cell, so a box of deltas is evaluated with one numpy broadcast.
"""

from __future__ import annotations

import math

import numpy as np

NGL_INT = 100  # hmf.c:87
EPS2 = 3.0e-11  # hmf.c:22
MAX_DELTAC_FRAC = float(np.float32(0.99))  # hmf.h:8
FRACT_FLOAT_ERR = 1e-7  # Constants.h
DELTA_C_SPH = 1.686  # Constants.c:42
JENKINS_A, JENKINS_B, JENKINS_C = 0.73, 0.34, 0.81  # hmf.c:48-50


def gauleg(x1: float, x2: float, n: int = NGL_INT):
    """hmf.c:659-694, statement for statement (1-based arrays upstream)."""
    x = np.zeros(n + 1)
    w = np.zeros(n + 1)
    m = (n + 1) // 2
    xm, xl = 0.5 * (x2 + x1), 0.5 * (x2 - x1)
    for i in range(1, m + 1):
        z = math.cos(3.141592654 * (i - 0.25) / (n + 0.5))
        while True:
            p1, p2 = 1.0, 0.0
            for j in range(1, n + 1):
                p3 = p2
                p2 = p1
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j
            pp = n * (z * p1 - p2) / (z * z - 1.0)
            z1 = z
            z = z1 - p1 / pp
            if not abs(z - z1) > EPS2:
                break
        x[i] = xm - xl * z
        x[n + 1 - i] = xm + xl * z
        w[i] = 2.0 * xl / ((1.0 - z * z) * pp * pp)
        w[n + 1 - i] = w[i]
    return x[1:], w[1:]


def sheth_delc_fixed(delta: float, sig: float) -> float:  # hmf.c:151-154
    return math.sqrt(JENKINS_A) * delta * (1.0 + JENKINS_B * (sig * sig / (JENKINS_A * delta * delta)) ** JENKINS_C)


def get_delta_crit_st(sigma: float, growthf: float) -> float:  # hmf.c:166-171, HMF = ST
    return sheth_delc_fixed(DELTA_C_SPH / growthf, sigma) * growthf


def st_taylor_factor(sig: np.ndarray, sig_cond: float, growthf: float):
    """hmf.c:234-267 for an array of sigmas; returns (factor, zeroth-order barrier)."""
    a, alpha, beta = JENKINS_A, JENKINS_C, JENKINS_B
    delta = DELTA_C_SPH / growthf
    sigsq = sig * sig
    sigsq_inv = 1.0 / sigsq
    sigdiff = np.where(sig == sig_cond, 1e-6, sigsq - sig_cond * sig_cond)
    t = [np.ones_like(sig)]
    for i in range(1, 6):
        t.append(t[i - 1] * (-sigdiff) / i * (alpha - i + 1) * sigsq_inv)
    result = np.zeros_like(sig)
    for i in range(5, -1, -1):  # "sum small to large"
        result = result + t[i]
    pre1 = math.sqrt(a) * delta
    pre2 = beta * (sigsq_inv * (a * delta * delta)) ** (-alpha)
    return pre1 * (1 + pre2 * result), pre1 * (1 + pre2)


M_MIN_INTEGRAL, M_MAX_INTEGRAL = 1e5, 1e16  # hmf.h:10-11


def mass_limit_bisection(PL: float, FRAC: float, Mmin: float = M_MIN_INTEGRAL, Mmax: float = M_MAX_INTEGRAL) -> float:
    """hmf.c:1268-1311: the mass above (below) which f_star / f_esc would exceed unity, found by a FLOAT
    bisection in log10 M that stops when two midpoints are 1e-3 apart (so it is NOT 1e10 FRAC^(-1/PL): the
    reference's own comment asks why) -- scaling_relations.c:106-109 stores it in the scaling constants."""
    f32 = np.float32

    def mass_limit(logM):  # float Mass_limit(float, float, float): double pow, float result
        return f32(float(f32(FRAC)) * (10.0 ** float(logM) / 1e10) ** float(f32(PL)))

    lo, up = f32(math.log10(Mmin)), f32(math.log10(Mmax))
    if PL < 0.0:
        if mass_limit(lo) <= 1.0:
            return float(f32(Mmin))
    elif PL > 0.0:
        if mass_limit(up) <= 1.0:
            return float(f32(Mmax))
    else:
        return 0.0
    x = f32((float(lo) + float(up)) / 2.0)
    for _ in range(200):
        if (float(mass_limit(lo)) - 1.0) * (float(mass_limit(x)) - 1.0) < 0.0:
            up = x
        else:
            lo = x
        x1 = f32((float(lo) + float(up)) / 2.0)
        if abs(float(x1) - float(x)) < float(f32(0.001)):
            return float(f32(10.0 ** float(x1)))
        x = x1
    raise RuntimeError("Mass_limit_bisection did not converge")


def log_scaling_pl_limit(lnM, ln_norm, alpha, ln_pivot, ln_limit):  # scaling_relations.c:225-231
    lim = ((alpha > 0.0) & (lnM > ln_limit)) | ((alpha < 0.0) & (lnM < ln_limit))
    return np.where(lim, -ln_norm, alpha * (lnM - ln_pivot))


def nion_fraction(lnM, sc: dict, mturn: float):  # hmf.c:462-468
    fstar = log_scaling_pl_limit(lnM, math.log(sc["fstar_10"]), sc["alpha_star"], 10 * math.log(10.0),
                                 math.log(sc["Mlim_Fstar"]))
    fesc = log_scaling_pl_limit(lnM, math.log(sc["fesc_10"]), sc["alpha_esc"], 10 * math.log(10.0),
                                math.log(sc["Mlim_Fesc"]))
    return np.exp(fstar + fesc - mturn / np.exp(lnM) + lnM)


class ConditionalNion:
    """Nion_ConditionalM (hmf.c:1106-1140) with the Gauss-Legendre rule for one (z, M_min, M_cond)."""

    def __init__(self, cosmo, growthf: float, M_min: float, M_cond: float, sc: dict, mturn: float,
                 sigma_cond: float | None = None):
        self.growthf = growthf
        self.lnM1, self.lnM2 = math.log(M_min), math.log(M_cond)
        self.sigma_cond = cosmo.sigma_z0(M_cond) if sigma_cond is None else sigma_cond
        self.sc, self.mturn = sc, mturn
        self.empty = self.lnM1 >= self.lnM2  # hmf.c:1123
        if self.empty:
            return
        self.x, self.w = gauleg(self.lnM1, self.lnM2)
        M = np.exp(self.x)
        self.sigma = np.array([cosmo.sigma_z0(m) for m in M])
        self.dsigmasqdm = np.array([cosmo.dsigmasqdm_z0(m) for m in M])
        self.nion = nion_fraction(self.x, sc, mturn)
        self.factor, self.barrier = st_taylor_factor(self.sigma, self.sigma_cond, growthf)
        s1, sc2 = self.sigma, self.sigma_cond
        self.sigdiff_inv = np.where(s1 == sc2, 1e6, 1.0 / (s1 * s1 - sc2 * sc2))
        self.live = s1 >= sc2  # hmf.c:275: zero below the condition's sigma
        self.delta_limit = MAX_DELTAC_FRAC * get_delta_crit_st(self.sigma_cond, growthf)
        one = nion_fraction(np.array([self.lnM2]), sc, mturn)[0] / math.exp(self.lnM2)
        self.collapsed = one if self.lnM2 * (1 - FRACT_FLOAT_ERR) <= self.lnM2 else 0.0

    def __call__(self, delta) -> np.ndarray:
        delta = np.asarray(delta, dtype=np.float64)
        if self.empty:
            return np.zeros_like(delta)
        d0 = (delta / self.growthf)[..., None]  # hmf.c:272
        f = (-self.dsigmasqdm * (self.factor - d0) * self.sigdiff_inv ** 1.5
             * np.exp(-(self.barrier - d0) ** 2 * 0.5 * self.sigdiff_inv) / math.sqrt(2.0 * math.pi))
        f = np.where(self.live, f, 0.0)
        integral = np.sum(self.w * self.nion * f, axis=-1)
        return np.where(delta > self.delta_limit, self.collapsed, integral)
