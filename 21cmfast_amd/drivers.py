"""Host-side mirror of the part of py21cmfast's driver layer that sits between the C entry points
of the spin-temperature path for the Lagrangian source models
(reference: src/py21cmfast/drivers/single_field.py:382-470 ``interp_halo_boxes`` and :473-636
``compute_xray_source_field``; the redshift loop that calls them is drivers/coeval.py:749-890).

The reference does this bookkeeping in Python with astropy: the shells of the X-ray / Lyman-alpha
light cone are placed in comoving distance, every shell takes the halo grids (``halo_sfr``,
``halo_xray``) linearly interpolated between the two snapshots that bracket its mean redshift,
and ``UpdateXraySourceBox`` filters them into ``XraySourceBox.filtered_sfr / filtered_xray``.
astropy is not a dependency here: the comoving distance of its ``FlatLambdaCDM`` (photons plus
three neutrino species, one of 0.06 eV, as in its Planck18 realisation) is restated with numpy.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import structs as S
from ._lib import check, load

L_FACTOR = (4 * math.pi / 3.0) ** (-1 / 3)  # single_field.py:516
C_KMS = 299792.458
MPC_CM = 3.085677581491367e24  # astropy's Mpc
G_CGS = 6.6743e-8              # CODATA 2018, as astropy.constants
M_P = 1.67262192369e-24
SIGMA_SB = 5.670374419e-5
C_CMS = 2.99792458e10
K_B_EV = 8.617333262e-5


class FlatCosmology:
    """astropy.cosmology.FlatLambdaCDM(H0, Om0, Ob0, Tcmb0 = 2.7255 K, Neff = 3.046,
    m_nu = [0, 0, 0.06] eV) -- what ``CosmoParams.cosmo`` is for the Planck18 base
    (reference: wrapper/inputs.py:603-611).  Only E(z) and the comoving distance are needed."""

    def __init__(self, hlittle, OMm, Tcmb0=2.7255, Neff=3.046, m_nu=(0.0, 0.0, 0.06)):
        self.h, self.Om0 = float(hlittle), float(OMm)
        self.H0_cgs = self.h * 100.0 * 1e5 / MPC_CM  # 1/s
        self.rho_crit0 = 3 * self.H0_cgs**2 / (8 * math.pi * G_CGS)  # g / cm^3
        a_rad = 4 * SIGMA_SB / C_CMS
        self.Ogamma0 = a_rad * Tcmb0**4 / (self.rho_crit0 * C_CMS**2)
        m = np.array(m_nu, float)
        self.n_massless = int(np.sum(m == 0))
        self.nu_y = m[m > 0] / (K_B_EV * 0.7137658555036082 * Tcmb0)
        self.neff_per_nu = Neff / 3.0
        self.Onu0 = self.Ogamma0 * self._nu_rel(0.0)
        self.Ode0 = 1.0 - self.Om0 - self.Ogamma0 - self.Onu0

    def _nu_rel(self, z):
        """Komatsu et al. 2011 eq. 26 as astropy evaluates it (nu_relative_density)."""
        p, invp, k = 1.83, 0.54644808743, 0.3173
        z = np.asarray(z, float)
        y = self.nu_y[None, :] / (1.0 + z[..., None]) if z.ndim else self.nu_y / (1.0 + z)
        rel = np.sum((1.0 + (k * y) ** p) ** invp, axis=-1) + self.n_massless
        return 0.22710731766 * self.neff_per_nu * rel

    def efunc(self, z):
        z = np.asarray(z, float)
        zp1 = 1.0 + z
        return np.sqrt(zp1**3 * (self.Ogamma0 * (1 + self._nu_rel(z)) * zp1 + self.Om0) + self.Ode0)

    def comoving_distance(self, z):
        """Mpc; composite Gauss-Legendre in ln(1 + z), converged to ~1e-10."""
        x, w = np.polynomial.legendre.leggauss(12)
        out = []
        for zz in np.atleast_1d(np.asarray(z, float)):
            edges = np.linspace(0.0, math.log1p(zz), 17)
            tot = 0.0
            for a, b in zip(edges[:-1], edges[1:]):
                u = 0.5 * (a + b) + 0.5 * (b - a) * x
                tot += 0.5 * (b - a) * float(np.sum(w * np.exp(u) / self.efunc(np.expm1(u))))
            out.append(C_KMS / (100.0 * self.h) * tot)
        return np.array(out) if np.ndim(z) else out[0]

    def z_at_comoving_distance(self, d):
        lo, hi = 0.0, 2000.0
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if self.comoving_distance(mid) < d:
                lo = mid
            else:
                hi = mid
            if hi - lo < 1e-12 * max(1.0, hi):
                break
        return 0.5 * (lo + hi)


def xray_shells(redshift, hii_dim, box_len, n_step, r_max_ts, cosmo: FlatCosmology):
    """Outer radii [Mpc] and mean redshifts of the shells (single_field.py:515-546): edges in
    comoving distance from `redshift`, converted with a 100-point log grid in z, mean = outer
    edge minus half the shell's redshift width."""
    R_min = (1.5 if hii_dim == 1 else box_len / hii_dim) * L_FACTOR
    steps = np.arange(0, n_step)
    R_range = R_min * (r_max_ts / R_min) ** (steps / n_step)
    cmd_edges = cosmo.comoving_distance(redshift) + R_range
    zmin = cosmo.z_at_comoving_distance(cmd_edges.min())
    zmax = cosmo.z_at_comoving_distance(cmd_edges.max())
    zgrid = np.logspace(np.log10(zmin), np.log10(zmax), 100)
    dgrid = cosmo.comoving_distance(zgrid)
    zpp_edges = np.interp(cmd_edges, dgrid, zgrid)
    zpp_avg = zpp_edges - np.diff(np.insert(zpp_edges, 0, redshift)) / 2
    return R_range, zpp_avg


def interp_halo_boxes(z_halos, boxes, fields, redshift):
    """Linear interpolation of the halo grids between the two snapshots bracketing `redshift`
    (single_field.py:382-470).  z_halos ascending; boxes: dicts of arrays.  Returns a dict."""
    z_halos = list(z_halos)
    if not np.all(np.diff(z_halos) > 0):
        raise ValueError("halo_boxes must be in ascending order of redshift")
    if redshift > z_halos[-1] or redshift < z_halos[0]:
        raise ValueError(f"Invalid z_target {redshift} for redshift array {z_halos}")
    idx_prog = int(np.searchsorted(z_halos, redshift, side="left"))
    if idx_prog == 0 or idx_prog == len(z_halos):
        raise ValueError(f"redshift {redshift} beyond limits {z_halos[0], z_halos[-1]}")
    idx_desc = idx_prog - 1
    t = (redshift - z_halos[idx_desc]) / (z_halos[idx_prog] - z_halos[idx_desc])
    out = {}
    for f in fields:
        interp = np.zeros_like(boxes[idx_desc][f])
        interp[...] = (1 - t) * boxes[idx_desc][f] + t * boxes[idx_prog][f]
        out[f] = interp
    return out


def lya_diffusion_scale(redshift, x_HI, hlittle, OMm, OMb, Y_He, cosmo: FlatCosmology):
    """R_star [Mpc] of the multiple-scattering window, eq. 24 of arXiv:2601.14360 as coded at
    single_field.py:549-572."""
    A_alpha, nu_Lya = 6.25e8, 2.46606727e15
    n_H_z0 = (1.0 - Y_He) * cosmo.rho_crit0 * OMb / M_P
    R = 3.0 * C_CMS**4 * A_alpha**2 * n_H_z0 * x_HI * (1.0 + redshift)
    R /= 32.0 * math.pi**3 * nu_Lya**4 * cosmo.H0_cgs**2 * OMm
    return R / MPC_CM


def compute_xray_source_field(z_halos, hboxes, redshift, *, simulation_options, cosmo_params,
                              astro_params, astro_options, previous_xHI_mean=None, lib=None):
    """XraySourceBox for `redshift` from the halo-grid history (compute_xray_source_field,
    single_field.py:473-636).  `z_halos` / `hboxes`: redshifts (descending, as the evolution
    produces them, the current one last) and dicts with ``halo_sfr`` and ``halo_xray`` grids.
    The process-global parameters must have been broadcast to the library.  Returns a dict with
    ``filtered_sfr``, ``filtered_xray`` [N_STEP_TS, ...] and ``mean_sfr`` [N_STEP_TS]."""
    lib = lib or load(require_gpu=True)
    so, cp, ap, ao = simulation_options, cosmo_params, astro_params, astro_options
    if ao.USE_MINI_HALOS:
        raise NotImplementedError("USE_MINI_HALOS is not supported by this backend")
    n_step = ap.N_STEP_TS
    shape = hboxes[0]["halo_sfr"].shape
    cosmo = FlatCosmology(cp.hlittle, cp.OMm)
    R_range, zpp_avg = xray_shells(redshift, so.HII_DIM, so.BOX_LEN, n_step, ap.R_MAX_TS, cosmo)
    z_max = min(max(z_halos), so.Z_HEAT_MAX)
    if ao.LYA_MULTIPLE_SCATTERING:
        x_HI = 1.0 if previous_xHI_mean is None else float(previous_xHI_mean)
        R_star = lya_diffusion_scale(redshift, x_HI, cp.hlittle, cp.OMm, cp.OMb, cp.Y_He, cosmo)
    else:
        R_star = 0.0
    box = {"filtered_sfr": np.zeros((n_step,) + shape, np.float32),
           "filtered_xray": np.zeros((n_step,) + shape, np.float32),
           "mean_sfr": np.zeros(n_step), "mean_sfr_mini": np.zeros(n_step),
           "mean_log10_Mcrit_LW": np.zeros(n_step)}
    src = S.XraySourceBoxStruct(
        filtered_sfr=box["filtered_sfr"].ctypes.data_as(S.c_float_p),
        filtered_xray=box["filtered_xray"].ctypes.data_as(S.c_float_p),
        mean_sfr=box["mean_sfr"].ctypes.data_as(C.POINTER(C.c_double)),
        mean_sfr_mini=box["mean_sfr_mini"].ctypes.data_as(C.POINTER(C.c_double)),
        mean_log10_Mcrit_LW=box["mean_log10_Mcrit_LW"].ctypes.data_as(C.POINTER(C.c_double)))
    order = np.argsort(z_halos)
    z_sorted = [z_halos[i] for i in order]
    b_sorted = [hboxes[i] for i in order]
    for i in range(n_step):
        R_inner = float(R_range[i - 1]) if i > 0 else 0.0
        R_outer = float(R_range[i])
        if zpp_avg[i] >= z_max:  # above Z_HEAT_MAX or the first snapshot: nothing shines yet
            continue
        hb = interp_halo_boxes(z_sorted, b_sorted, ("halo_sfr", "halo_xray"), float(zpp_avg[i]))
        if np.all(hb["halo_sfr"] == 0):
            continue
        hbs = S.HaloBoxStruct(halo_sfr=hb["halo_sfr"].ctypes.data_as(S.c_float_p),
                              halo_xray=hb["halo_xray"].ctypes.data_as(S.c_float_p))
        check(lib.UpdateXraySourceBox(C.byref(hbs), R_inner, R_outer, i, R_star, C.byref(src)),
              "UpdateXraySourceBox")
    box["zpp_avg"], box["R_range"], box["R_star"] = zpp_avg, R_range, R_star
    return box
