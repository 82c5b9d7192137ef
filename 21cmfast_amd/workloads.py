"""Synthetic inputs and explicit-scalar specs for the IonizedBox hot path.

These are the benchmark / parity workloads of SURVEY.md section 8(d): a Gaussian random
density field (P(k) ~ k^-2, sigma = 0.25, clipped at -0.95), an emissivity grid
``n_ion = rho_crit * Omega_b * fbar * (1+delta)^2.5`` and the radius ladder that
``setup_radii`` (reference: src/py21cmfast/src/IonisationBox.c:964-1006) produces for
``R_BUBBLE_MAX = 40`` on 1.5 Mpc cells: 40 radii from 0.9305 to 38.29 Mpc.

Only numpy / torch array generation and struct filling happen here; no physics is
computed on the host.
"""

from __future__ import annotations

import math

import numpy as np

from . import structs as S

L_FACTOR = 0.620350491  # reference: src/py21cmfast/src/Constants.c:41
DELTA_C = 1.686  # reference: Constants.c:42
FCOLL_STARS, FCOLL_ERFC, FCOLL_TABLE_LINEAR, FCOLL_TABLE_EXP = 0, 1, 2, 3

# The reference's Planck18 (inputs.py:126-134: Om0 = (0.02242 + 0.11933) / h^2, Ob0 = 0.02242 / h^2)
HLITTLE = 0.6766
OMM, OMB = (0.02242 + 0.11933) / HLITTLE**2, 0.02242 / HLITTLE**2


def rho_crit(hlittle: float = HLITTLE) -> float:
    """Critical density in Msun / Mpc^3 (reference: src/py21cmfast/src/Constants.h RHOcrit)."""
    Ho = float(np.float32(hlittle)) * 3.2407e-18
    G, cm_per_Mpc, Msun = 6.6743e-8, 3.08567758e24, 1.989e33
    return float((3.0 * Ho * Ho / (8.0 * math.pi * G)) * cm_per_Mpc**3 / Msun)


def radii_ladder(hii_dim: int, box_len: float, r_bubble_max: float = 40.0,
                 r_bubble_min: float = L_FACTOR, delta_r_factor: float = 1.1,
                 lagrangian: bool = True) -> list[float]:
    """Filter radii exactly as ``setup_radii`` builds them (IonisationBox.c:964-1006)."""
    box_len = float(np.float32(box_len))
    maximum_radius = min(float(np.float32(r_bubble_max)), L_FACTOR * box_len)
    pixel_length = box_len / float(hii_dim)
    cell_length_factor = L_FACTOR
    if lagrangian and pixel_length < 1:
        cell_length_factor = 1.0
    minimum_radius = max(float(np.float32(r_bubble_min)), cell_length_factor * pixel_length)
    n_radii = int(math.log(maximum_radius / minimum_radius) / math.log(delta_r_factor) + 1)
    radii = []
    for i in range(n_radii):
        R = minimum_radius * delta_r_factor**i
        if R > maximum_radius - 1e-7:
            radii.append(maximum_radius)
            break
        radii.append(R)
    return radii


def ionize_spec(hii_dim: int, box_len: float | None = None, mode: int = FCOLL_STARS,
                r_bubble_max: float = 40.0, redshift: float = 9.0, hii_dim_z: int | None = None,
                **overrides) -> S.IonizeSpec:
    """Spec of benchmark config 3 (G = 2, L-INTEGRAL semantics) or its G = 1 variant."""
    if box_len is None:
        box_len = 1.5 * hii_dim
    lagrangian = mode == FCOLL_STARS
    spec = S.IonizeSpec()
    spec.hii_dim = hii_dim
    spec.hii_dim_z = hii_dim if hii_dim_z is None else hii_dim_z
    spec.box_len = float(np.float32(box_len))
    spec.box_len_z = spec.box_len * spec.hii_dim_z / hii_dim
    radii = radii_ladder(hii_dim, box_len, r_bubble_max, lagrangian=lagrangian)
    spec.n_radii = len(radii)
    spec.r_lowest = 0
    for i, R in enumerate(radii):
        spec.R[i] = R
        # sigma(M(R)) stand-in, monotonically decreasing with R (ERFC / TABLE_LINEAR modes);
        # the real value comes from sigma_z0(RtoM(R)) in ComputeIonizedBox
        spec.sigma_maxmass[i] = 2.4 * (R / 0.6) ** -0.62
    if lagrangian:
        spec.hii_filter = 0  # real-space top-hat
        spec.stars_filter = 3  # USE_EXP_FILTER
    else:
        spec.hii_filter = 1  # sharp-k ("simple" / "const-zeta" templates)
        spec.stars_filter = 1
    spec.mfp_meandens = 25.483241248322766 / float(np.float32(HLITTLE))
    spec.fcoll_mode = mode
    spec.fix_mean = 0 if lagrangian else 1
    spec.mass_dep_zeta = 1 if lagrangian else 0
    spec.use_ts_fluct = 0
    spec.recomb_model = 0
    spec.cell_recomb = 1
    spec.minimize_memory = 0
    spec.first_snapshot = 1
    spec.redshift = float(np.float32(redshift))
    spec.stored_redshift = spec.redshift
    spec.photoncons_adjustment_factor = 1.0
    spec.ion_eff_factor = 1.0 if lagrangian else 30.0
    spec.mean_f_coll = 0.5 if lagrangian else 0.5 / 30.0
    spec.f_limit_acg = 1e-9
    spec.gamma_prefactor = 1.0
    spec.rhocrit_omb = rho_crit() * float(np.float32(OMB))
    spec.growth_factor = 0.1266  # ~ D(z=9)
    spec.sigma_minmass = 4.5     # ~ sigma(M_min ~ 1e8 Msun)
    spec.delta_c = DELTA_C
    spec.TK_nofluct = 2.5
    spec.adia_TK_term = 0.6
    spec.T_re = 2e4
    spec.fabs_dtdz = 0.0
    spec.dz = 0.0
    for k, v in overrides.items():
        setattr(spec, k, v)
    return spec


def density_field_numpy(shape, seed: int = 12345, sigma: float = 0.25) -> np.ndarray:
    """Gaussian random field with P(k) ~ k^-2, normalised to `sigma`, clipped at -0.95."""
    if isinstance(shape, int):
        shape = (shape,) * 3
    rng = np.random.default_rng(seed)
    white = rng.standard_normal(shape).astype(np.float32)
    wk = np.fft.rfftn(white)
    kx = np.fft.fftfreq(shape[0])[:, None, None]
    ky = np.fft.fftfreq(shape[1])[None, :, None]
    kz = np.fft.rfftfreq(shape[2])[None, None, :]
    k2 = kx * kx + ky * ky + kz * kz
    k2[0, 0, 0] = 1.0
    wk *= 1.0 / np.sqrt(k2)  # amplitude ~ k^-1  ->  P ~ k^-2
    wk[0, 0, 0] = 0.0
    field = np.fft.irfftn(wk, s=shape, axes=(0, 1, 2)).astype(np.float32)
    field *= sigma / field.std()
    np.maximum(field, -0.95, out=field)
    return np.ascontiguousarray(field, np.float32)


def density_field_torch(n: int, seed: int = 12345, sigma: float = 0.25, device="cuda"):
    """Same construction on the GPU (used by bench.py at 512^3 and above)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    white = torch.randn((n, n, n), generator=g, device=device, dtype=torch.float32)
    wk = torch.fft.rfftn(white)
    del white
    f = torch.fft.fftfreq(n, device=device)
    fz = torch.fft.rfftfreq(n, device=device)
    k2 = f[:, None, None] ** 2 + f[None, :, None] ** 2 + fz[None, None, :] ** 2
    k2[0, 0, 0] = 1.0
    wk *= torch.rsqrt(k2)
    wk[0, 0, 0] = 0.0
    del k2
    field = torch.fft.irfftn(wk, s=(n, n, n))
    del wk
    field *= sigma / field.std()
    field.clamp_(min=-0.95)
    return field.contiguous()


def nion_from_density(density, fbar: float = 0.9, hlittle: float = HLITTLE, omb: float = OMB):
    """``HaloBox.n_ion`` stand-in: rho_crit*Omega_b*fbar*(1+delta)^2.5 (numpy or torch)."""
    norm = rho_crit(hlittle) * float(np.float32(omb)) * fbar
    out = (1.0 + density) ** 2.5 * norm
    if isinstance(out, np.ndarray):
        return np.ascontiguousarray(out, np.float32)
    return out.float().contiguous()
