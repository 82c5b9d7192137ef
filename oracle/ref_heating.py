"""Host scalars of the spin-temperature calculation evaluated INDEPENDENTLY of
lib21cmfast_hip.so (numpy / scipy).  TEST INFRASTRUCTURE ONLY.

Restates, formula by formula, what ComputeTsBox prepares on the host before its cell loops
(reference: src/py21cmfast/src/SpinTemperatureBox.c and heating_helper_progs.c), so that the
tests can drive the oracle with numbers the library under test did not produce, and check the
library's own tables against them.

reference lines followed:
  Constants.c:4-45, Constants.h:90-112        physical constants, No, He_No, N_b0, H_FRAC, HE_FRAC
  SpinTemperatureBox.c:312-361                setup_z_edges
  SpinTemperatureBox.c:364-499                calculate_spectral_factors
  SpinTemperatureBox.c:1098-1184              set_zp_consts
  heating_helper_progs.c:195-264              frecycle
  heating_helper_progs.c:268-353              spectral_emissivity (stellar_spectra.dat)
  heating_helper_progs.c:356-362              nu_n;  :1193-1198 zmax
  heating_helper_progs.c:767-858              the three frequency integrands, integrate_over_nu
  heating_helper_progs.c:862-872              species_weighted_x_ray_cross_section
  heating_helper_progs.c:943-1059             tauX_integrand, tauX
  heating_helper_progs.c:1135-1190            nu_tau_one (Brent root of tauX = 1)
  elec_interp.c:39-115,117-423                x_int tables and their bilinear lookups
  thermochem.c:104-146                        photo-ionisation cross sections
  cosmology.c:716-735                         dtdz, drdz, ddicke_dz
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from .ref_scalars import CM_PER_MPC, Cosmo, _f32

PC = dict(  # Constants.c:4-45
    c_cms=2.99792458e10, h_p=6.62607015e-27, k_B=1.380649e-16, m_p=1.67262192369e-24,
    m_e=9.1093837015e-28, G=6.6743e-8, e_charge=4.803204712570263e-10, vac_perm=8.8541878128e-12,
    Msun=1.989e33, s_per_yr=31556925.9747, cm_per_Mpc=CM_PER_MPC, eV_to_Hz=2.417989e14,
    nu_ion_HI=3.288465e15, nu_ion_HeI=5.945836e15, nu_ion_HeII=1.3153862e16,
    nu_LW_thresh=2.70331197e15, nu_Ly_alpha=2.46606727e15, T_cmb=2.7255, T_21=0.0682,
    lambda_21=21.106114054160, lambda_Ly_alpha=1215.67, A10=2.85e-15, f_alpha=0.4162,
    l_factor=0.620350491,
)
NSPEC_MAX = 23
X_INT_XHII = np.array([1.0e-4, 2.318e-4, 4.677e-4, 1.0e-3, 2.318e-3, 4.677e-3, 1.0e-2, 2.318e-2,
                       4.677e-2, 1.0e-1, 0.5, 0.9, 0.99, 0.999], np.float32)
X_INT_NENERGY = 258


@dataclass
class Densities:
    """Constants.h:98-112 for a cosmology (Y_He is a float in CosmoParams)."""

    No: float
    He_No: float
    N_b0: float
    h_frac: float
    he_frac: float


def densities(c: Cosmo, Y_He: float = 0.24) -> Densities:
    y = _f32(Y_He)
    Ho = c.h * 3.2407e-18
    rho_cgs = 3.0 * Ho * Ho / (8.0 * math.pi * PC["G"])
    No = rho_cgs * c.ob * (1 - y) / PC["m_p"]
    He = rho_cgs * c.ob * y / (4.0 * PC["m_p"])
    return Densities(No, He, No + He, (1.0 - y) / (1.0 - 3.0 * y / 4.0), (y / 4.0) / (1.0 - 3.0 * y / 4.0))


def drdz(c: Cosmo, z: float) -> float:  # cosmology.c:778-779: (1 + z) c dtdz, float argument
    z = _f32(z)
    return (1.0 + z) * PC["c_cms"] * c.dtdz(z)


def ddicke_dz(c: Cosmo, z: float) -> float:  # cosmology.c:586-590, float dz
    dz = _f32(1e-10)
    return (c.dicke(z + dz) - c.dicke(z)) / dz


def zp_consts(c: Cosmo, zp: float, *, lagrangian: bool, X_RAY_SPEC_INDEX=1.0, NU_X_THRESH=500.0,
              NU_X_BAND_MAX=2000.0, Y_He=0.24) -> dict:
    """set_zp_consts (SpinTemperatureBox.c:1098-1184); keys are c21cm_ts_spec's field names."""
    d = densities(c, Y_He)
    eV = PC["eV_to_Hz"]
    hub = c.hubble(zp)
    Ho = c.h * 3.2407e-18
    if abs(X_RAY_SPEC_INDEX - 1.0) < 1e-6:
        lum = 1.0 / (NU_X_THRESH * eV * math.log(NU_X_BAND_MAX / NU_X_THRESH))
    else:
        lum = 1.0 / ((NU_X_BAND_MAX * eV) ** (1.0 - X_RAY_SPEC_INDEX)
                     - (NU_X_THRESH * eV) ** (1.0 - X_RAY_SPEC_INDEX))
        lum *= (NU_X_THRESH * eV) ** (-X_RAY_SPEC_INDEX) * (1 - X_RAY_SPEC_INDEX)
    lum /= PC["h_p"]
    Trad = PC["T_cmb"] * (1.0 + zp)
    gamma_alpha = PC["f_alpha"] * (PC["nu_Ly_alpha"] * PC["e_charge"] / (PC["c_cms"] / 10.0)) ** 2
    gamma_alpha /= 6.0 * (PC["m_e"] / 1000.0) * (PC["c_cms"] / 100.0) ** 3 * PC["vac_perm"]
    xa = 8.0 * math.pi * (PC["lambda_Ly_alpha"] * 1e-8) ** 2 * gamma_alpha * PC["T_21"]
    xa /= 9.0 * PC["A10"] * Trad
    rhocrit = c.rhocrit()
    return dict(
        xray_prefactor=lum / (NU_X_THRESH * eV) * PC["c_cms"] * (1 + zp) ** (X_RAY_SPEC_INDEX + 3),
        Trad=Trad,
        Ts_prefactor=(1.0e-7 * (1.342881e-7 / hub) * d.No * (1 + zp) ** 3) ** (1.0 / 3.0),
        xa_tilde_prefactor=xa,
        xc_inverse=(1.0 + zp) ** 3 * PC["T_21"] / (Trad * PC["A10"]),
        dcomp_dzp_prefactor=(-1.51e-4) / (hub / Ho) / c.h * Trad**4 / (1.0 + zp),
        Nb_zp=d.N_b0 * (1 + zp) ** 3,
        N_zp=d.No * (1 + zp) ** 3,
        lya_star_prefactor=PC["c_cms"] / (4.0 * math.pi) * PC["Msun"] / PC["m_p"] * (1 - 0.75 * _f32(Y_He)),
        volunit_inv=(PC["cm_per_Mpc"] ** -3 if lagrangian else c.ob * rhocrit * PC["cm_per_Mpc"] ** -3),
        hubble_zp=hub,
        growth_zp=c.dicke(zp),
        dgrowth_dzp=ddicke_dz(c, zp),
        dt_dzp=c.dtdz(zp),
        No=d.No, N_b0=d.N_b0, h_frac=d.h_frac, he_frac=d.he_frac,
        k_B=PC["k_B"], h_p=PC["h_p"], m_p=PC["m_p"], c_cms=PC["c_cms"], A10=PC["A10"],
        T_21=PC["T_21"], lambda_21=PC["lambda_21"], nu_Ly_alpha=PC["nu_Ly_alpha"],
    )


# ------------------------------------------------------------------ shells (setup_z_edges)
def z_edges(c: Cosmo, zp: float, hii_dim: int, box_len: float, n_step=40, R_MAX_TS=500.0) -> dict:
    """setup_z_edges (:312-361) without the mass limits: R_values, zpp_edge, zpp (shell centre),
    dzpp, dtdz(zpp), growth(zpp)."""
    R = PC["l_factor"] * box_len / _f32(hii_dim)
    R_factor = (R_MAX_TS / R) ** (1 / _f32(n_step))
    out = {k: np.zeros(n_step) for k in ("R", "zpp_edge", "zpp", "dzpp", "dtdz", "growth")}
    for i in range(n_step):
        out["R"][i] = R
        prev_zpp = zp if i == 0 else out["zpp_edge"][i - 1]
        prev_R = 0.0 if i == 0 else out["R"][i - 1]
        out["zpp_edge"][i] = prev_zpp - (R - prev_R) * PC["cm_per_Mpc"] / drdz(c, prev_zpp)
        out["zpp"][i] = (out["zpp_edge"][i] + prev_zpp) * 0.5
        out["dzpp"][i] = (zp - out["zpp_edge"][0]) if i == 0 else (out["zpp_edge"][i - 1] - out["zpp_edge"][i])
        out["growth"][i] = c.dicke(out["zpp"][i])
        out["dtdz"][i] = c.dtdz(out["zpp"][i])
        R *= R_factor
    return out


# ------------------------------------------------------------------ stellar spectra
def frecycle(n: int) -> float:  # heating_helper_progs.c:195-264 (Pritchard & Furlanetto 2006)
    tab = {0: 1, 1: 1, 2: 1, 3: 0, 4: 0.2609, 5: 0.3078, 6: 0.3259, 7: 0.3353, 8: 0.3410,
           9: 0.3448, 10: 0.3476, 11: 0.3496, 12: 0.3512, 13: 0.3524, 14: 0.3535, 15: 0.3543,
           16: 0.3550, 17: 0.3556, 18: 0.3561, 19: 0.3565, 20: 0.3569, 21: 0.3572, 22: 0.3575,
           23: 0.3578, 24: 0.3580, 25: 0.3582, 26: 0.3584, 27: 0.3586, 28: 0.3587, 29: 0.3589,
           30: 0.3590}
    return float(tab.get(n, 0))


def nu_n(n: int) -> float:
    return (1.0 - float(n) ** -2.0) / 0.75


def zmax(z: float, n: int) -> float:  # returns float upstream
    num, den = 1 - float(n + 1) ** -2, 1 - float(n) ** -2
    return _f32((1 + _f32(z)) * num / den - 1)


class StellarSpectra:
    """spectral_emissivity (:268-353): piecewise power laws between the Lyman lines, read from
    stellar_spectra.dat (n, N0_II, alpha_II, N0_III, alpha_III per line, floats)."""

    def __init__(self, path, POP2_ION=5000.0, POP3_ION=44021.0):
        with open(path) as fh:  # 22 numeric rows (n = 2 .. 23), then a column-title line
            rows = np.array([ln.split() for ln in fh.read().splitlines()[: NSPEC_MAX - 1]], float)
        assert rows.shape == (NSPEC_MAX - 1, 5)
        self.n = np.zeros(NSPEC_MAX, int)
        self.N0 = {2: np.zeros(NSPEC_MAX, np.float32), 3: np.zeros(NSPEC_MAX, np.float32)}
        self.alpha = {2: np.zeros(NSPEC_MAX, np.float32), 3: np.zeros(NSPEC_MAX, np.float32)}
        for i in range(1, NSPEC_MAX):
            r = rows[i - 1]
            self.n[i] = int(r[0])
            self.N0[2][i], self.alpha[2][i], self.N0[3][i], self.alpha[3][i] = r[1:5]
        self.nu = np.zeros(NSPEC_MAX, np.float32)
        for i in range(1, NSPEC_MAX):
            self.nu[i] = 4.0 / 3.0 * (1.0 - 1.0 / float(self.n[i]) ** 2)
        for pop, ion in ((2, _f32(POP2_ION)), (3, _f32(POP3_ION))):
            for i in range(1, NSPEC_MAX - 1):
                a1 = float(self.alpha[pop][i]) + 1
                fac = float(self.nu[i + 1]) ** a1 - float(self.nu[i]) ** a1
                self.N0[pop][i] = np.float32(float(self.N0[pop][i]) * a1 / fac * ion)

    def emissivity(self, nu_norm: float, pop: int = 2) -> float:
        for i in range(1, NSPEC_MAX - 1):
            if self.nu[i] <= nu_norm < self.nu[i + 1]:
                return float(self.N0[pop][i]) * nu_norm ** float(self.alpha[pop][i]) / PC["nu_Ly_alpha"]
        i = NSPEC_MAX - 1
        return float(self.N0[pop][i]) * nu_norm ** float(self.alpha[pop][i]) / PC["nu_Ly_alpha"]

    def emissivity_lw(self, nu_norm: float, pop: int = 2) -> float:
        """spectral_emissivity(nu, 2, pop) (:284-302): photons between nu and the next Lyman
        line; outside the tabulated bands the C code falls through and returns 0"""
        for i in range(1, NSPEC_MAX - 1):
            if self.nu[i] <= nu_norm < self.nu[i + 1]:
                a1 = float(self.alpha[pop][i]) + 1
                r = float(self.N0[pop][i]) / a1 * (float(self.nu[i + 1]) ** a1 - nu_norm ** a1)
                return r if r > 0 else 1e-40
        return 0.0


def spectral_factors(spec: StellarSpectra, zp: float, zpp_list) -> dict:
    """calculate_spectral_factors (:364-499) without mini-halos: dstarlya_dt_prefactor and its
    continuum (n = 2) / injected (n > 2) parts per shell."""
    n = len(zpp_list)
    out = {k: np.zeros(n) for k in ("starlya", "cont", "inj")}
    first_radii, first_zero = True, True
    weight = 0.0
    prev = dict(lyn=0.0, ly2=0.0, lynto2=0.0)
    prev_zpp = 0.0
    for R_ct in range(n):
        zpp = float(zpp_list[R_ct])
        sum_ly2 = sum_lynto2 = 0.0
        nuprime = nu_n(2) * (1.0 + zpp) / (1.0 + zp)
        if zpp < zmax(zp, 2):
            sum_ly2 = frecycle(2) * spec.emissivity(nuprime, 2)
        for n_ct in range(NSPEC_MAX, 2, -1):
            if zpp > zmax(zp, n_ct):
                continue
            nuprime = nu_n(n_ct) * (1 + zpp) / (1.0 + zp)
            sum_lynto2 += frecycle(n_ct) * spec.emissivity(nuprime, 2)
        sum_lyn = sum_ly2 + sum_lynto2
        if R_ct > 1 and sum_lyn == 0.0 and prev["lyn"] > 0.0 and first_radii:
            n_pts = 1000
            for ii in range(n_pts):
                trial = prev_zpp + (zpp - prev_zpp) * _f32(ii) / (_f32(n_pts) - 1.0)
                counter = sum(1 for n_ct in range(NSPEC_MAX, 1, -1) if not trial > zmax(zp, n_ct))
                if counter == 0 and first_zero:
                    first_zero = False
                    weight = _f32(ii) / _f32(n_pts)
            sum_lyn, sum_ly2, sum_lynto2 = weight * prev["lyn"], weight * prev["ly2"], weight * prev["lynto2"]
            first_radii = False
        integrand = (1 + zp) ** 2 * (1 + zpp)
        out["starlya"][R_ct] = integrand * sum_lyn
        out["cont"][R_ct] = integrand * sum_ly2
        out["inj"][R_ct] = integrand * sum_lynto2
        prev = dict(lyn=sum_lyn, ly2=sum_ly2, lynto2=sum_lynto2)
        prev_zpp = zpp
    return out


# ------------------------------------------------------------------ x_int tables
class XIntTables:
    """elec_interp.c: the 14 secondary-ionisation tables (Furlanetto & Stoever 2010), floats."""

    FIELDS = ("fheat", "n_Lya", "nion_HI", "nion_HeI", "nion_HeII")

    def __init__(self, directory):
        d = Path(directory)
        self.energy = None
        self.tab = {k: np.zeros((len(X_INT_XHII), X_INT_NENERGY), np.float32) for k in self.FIELDS}
        for i, x in enumerate(X_INT_XHII):
            name = (f"log_xi_{math.log10(float(x)):1.1f}.dat" if x < 0.3 else f"xi_{float(x):1.3f}.dat")
            with open(d / name) as fh:
                lines = fh.read().splitlines()
            body = [ln.split() for ln in lines[3:3 + X_INT_NENERGY]]
            # skipline(1); one header row of 5 numbers (rest of that line is consumed by
            # skipline(2) together with the column-title line); then NENERGY rows of 9 numbers
            arr = np.array(body, dtype=np.float64)
            assert arr.shape == (X_INT_NENERGY, 9), (name, arr.shape)
            if self.energy is None:
                self.energy = arr[:, 0].astype(np.float32)
            self.tab["fheat"][i] = arr[:, 2]
            self.tab["n_Lya"][i] = arr[:, 4]
            self.tab["nion_HI"][i] = arr[:, 5]
            self.tab["nion_HeI"][i] = arr[:, 6]
            self.tab["nion_HeII"][i] = arr[:, 7]

    @staticmethod
    def locate_energy_index(En):  # :401-411
        En = np.float32(En)
        if En < 1008.88:
            return int(math.log(float(En) / 10.0) / 1.98026273e-2)
        return 232 + int(math.log(float(En) / 1008.88) / 9.53101798e-2)

    @staticmethod
    def locate_xHII_index(x):
        m = len(X_INT_XHII) - 1
        while np.float32(x) < X_INT_XHII[m]:
            m -= 1
        return m

    def interp(self, field: str, En: float, xHII: float) -> float:
        """interp_fheat & co (:117-399): float arithmetic, clamped arguments."""
        f32 = np.float32
        En, x = f32(En), f32(xHII)
        E = self.energy
        if En > 0.999 * float(E[-1]):
            En = f32(float(E[-1]) * 0.999)
        elif En < E[0]:
            return 1.0 if field == "fheat" else 0.0
        if x > float(X_INT_XHII[-1]) * 0.999:
            x = f32(float(X_INT_XHII[-1]) * 0.999)
        elif x < X_INT_XHII[0]:
            x = f32(1.001 * float(X_INT_XHII[0]))
        nl = self.locate_energy_index(En)
        nh = nl + 1
        ml = self.locate_xHII_index(x)
        mh = ml + 1
        t = self.tab[field]

        def along_E(m):
            r = f32((t[m][nh] - t[m][nl]) / (E[nh] - E[nl]))
            r = f32(r * f32(En - E[nl]))
            return f32(r + t[m][nl])

        lo, hi = along_E(ml), along_E(mh)
        r = f32(f32(hi - lo) / f32(X_INT_XHII[mh] - X_INT_XHII[ml]))
        r = f32(r * f32(x - X_INT_XHII[ml]))
        return float(f32(r + lo))


# ------------------------------------------------------------------ cross sections
def HI_ion_crosssec(nu):  # thermochem.c:133-146
    nu0 = PC["nu_ion_HI"]
    if nu < nu0:
        return 0.0
    if nu == nu0:
        nu += 1e-30
    eps = math.sqrt(nu / nu0 - 1)
    if eps == 0:
        return 6.3e-18
    return 6.3e-18 * (nu0 / nu) ** 4 * math.exp(4 - (4 * math.atan(eps) / eps)) / (1 - math.exp(-2 * math.pi / eps))


def HeII_ion_crosssec(nu):  # :118-131
    nu0 = PC["nu_ion_HeII"]
    if nu < nu0:
        return 0.0
    if nu == nu0:
        nu += 1e-30
    eps = math.sqrt(nu / nu0 - 1)
    if eps == 0:
        return 6.3e-18 / 4
    return 6.3e-18 / 4 * (nu0 / nu) ** 4 * math.exp(4 - (4 * math.atan(eps) / eps)) / (1 - math.exp(-2 * math.pi / eps))


def HeI_ion_crosssec(nu):  # :104-116 (Verner et al. 1996)
    if nu < PC["nu_ion_HeI"]:
        return 0.0
    x = nu / PC["eV_to_Hz"] / 13.61 - 0.4434
    y = math.sqrt(x * x + 2.136**2)
    return 9.492e-16 * ((x - 1) ** 2 + 2.039**2) * y ** (0.5 * 3.188 - 5.5) * (1.0 + math.sqrt(y / 1.469)) ** -3.188


def weighted_cross_section(nu, x_e, d: Densities):  # heating_helper_progs.c:862-872
    return (d.h_frac * (1 - x_e) * HI_ion_crosssec(nu) + d.he_frac * (1 - x_e) * HeI_ion_crosssec(nu)
            + d.he_frac * x_e * HeII_ion_crosssec(nu))


# ------------------------------------------------------------------ frequency integrals
def nu_integrand(tables: XIntTables, d: Densities, nu: float, x_e: float, flag: int,
                 NU_X_THRESH=500.0, X_RAY_SPEC_INDEX=1.0) -> float:
    """integrand_in_nu_{heat,ion,lya}_integral (:767-829); x_e is rounded to float upstream."""
    x_e = _f32(x_e)
    eV = PC["eV_to_Hz"]
    thr = ((PC["nu_ion_HI"], d.h_frac * (1 - x_e), HI_ion_crosssec),
           (PC["nu_ion_HeI"], d.he_frac * (1 - x_e), HeI_ion_crosssec),
           (PC["nu_ion_HeII"], d.he_frac * x_e, HeII_ion_crosssec))
    s = 0.0
    for nu0, frac, sigma in thr:
        E = (nu - nu0) / eV
        if flag == 0:
            s += tables.interp("fheat", E, x_e) * PC["h_p"] * (nu - nu0) * frac * sigma(nu)
        elif flag == 1:
            F = (tables.interp("nion_HI", E, x_e) + tables.interp("nion_HeI", E, x_e)
                 + tables.interp("nion_HeII", E, x_e) + 1)
            s += F * frac * sigma(nu)
        else:
            s += tables.interp("n_Lya", E, x_e) * frac * sigma(nu)
    return s * (nu / (NU_X_THRESH * eV)) ** (-X_RAY_SPEC_INDEX - 1)


def integrate_over_nu(tables, d, c: Cosmo, zp, x_e, lower, flag, NU_X_MAX=10000.0, panels=150,
                      **kw) -> float:
    """integrate_over_nu (:831-858).  The reference stops GSL's adaptive 15-point rule at a
    relative error ESTIMATE of 1 %; this is the converged integral: composite 8-point
    Gauss-Legendre over `panels` logarithmic panels (the integrand is piecewise smooth between the
    258 energy knots of the tables, which an adaptive rule started on the whole range can miss)."""
    upper = NU_X_MAX * PC["eV_to_Hz"]
    x, w = np.polynomial.legendre.leggauss(8)
    edges = np.linspace(math.log(lower), math.log(upper), panels + 1)
    val = 0.0
    for a, b in zip(edges[:-1], edges[1:]):
        lnnu = 0.5 * (a + b) + 0.5 * (b - a) * x
        nu = np.exp(lnnu)
        f = np.array([nu_integrand(tables, d, float(v), x_e, flag, **kw) for v in nu])
        val += 0.5 * (b - a) * float(np.sum(w * f * nu))
    if flag == 2:
        return val * PC["c_cms"] / (4.0 * math.pi) / PC["nu_Ly_alpha"] / c.hubble(zp)
    return val


# ------------------------------------------------------------------ tauX and the tau = 1 frequency
def tauX(c: Cosmo, d: Densities, nu, x_e, x_e_ave, zp, zpp, ion_eff, nion_of_z) -> float:
    """tauX (:1007-1059) for the mass-dependent source models: optical depth of the IGM between
    zpp and zp for a photon received at frequency nu.  nion_of_z: the global collapsed ionising
    fraction (EvaluateNionTs).  Converged quadrature (the reference stops at a 0.5 % estimate)."""
    from scipy import integrate

    nu_0 = nu / (1 + zp)

    def f(zhat):
        drpropdz = PC["c_cms"] * c.dtdz(zhat)
        n = d.N_b0 * (1 + zhat) ** 3
        fcoll = nion_of_z(zhat)
        fill = 1.0 if fcoll < 1e-20 else 1 - ion_eff * fcoll / (1.0 - x_e_ave)
        fill = max(fill, 1e-4)
        return drpropdz * n * fill * weighted_cross_section(nu_0 * (1 + zhat), x_e, d)

    val, _ = integrate.quad(f, zpp, zp, epsrel=1e-8, limit=500)
    return val


def nu_tau_one(c, d, zp, zpp, x_e, ion_eff, nion_of_z, NU_X_THRESH=500.0) -> float:
    """nu_tau_one (:1135-1190): frequency at which tauX = 1, not below the HeI edge."""
    from scipy import optimize

    if x_e > 0.9999:
        return NU_X_THRESH
    lo = PC["nu_ion_HeI"]
    if tauX(c, d, lo, x_e, x_e, zp, zpp, ion_eff, nion_of_z) < 1:
        return lo
    return optimize.brentq(lambda nu: tauX(c, d, nu, x_e, x_e, zp, zpp, ion_eff, nion_of_z) - 1, lo,
                           1e6 * PC["eV_to_Hz"], rtol=1e-10)
