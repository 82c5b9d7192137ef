"""Host-side mirror of py21cmfast's driver layer for the boxes this backend computes: the evolution
loop of ``run_coeval`` (reference: src/py21cmfast/drivers/coeval.py:560-890) as ``run_coeval`` /
``Inputs`` below, and the bookkeeping that sits between the C entry points of the
spin-temperature path for the Lagrangian source models
(reference: src/py21cmfast/drivers/single_field.py:382-470 ``interp_halo_boxes`` and :473-636
``compute_xray_source_field``).  A py21cmfast installation keeps using its own drivers on top of
the library; this module is for callers without it (tests, tools, stand-alone runs).

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
        if np.ndim(boxes[idx_desc][f]) == 0:  # box-level scalars (log10_Mcrit_MCG_ave)
            out[f] = (1 - t) * boxes[idx_desc][f] + t * boxes[idx_prog][f]
            continue
        a, b = boxes[idx_desc][f], boxes[idx_prog][f]
        if hasattr(a, "is_cuda"):  # torch tensors: the history stays in HBM
            out[f] = ((1 - t) * a + t * b).contiguous()
            continue
        interp = np.zeros_like(a)
        interp[...] = (1 - t) * a + t * b
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
    produces them, the current one last) and dicts with ``halo_sfr`` and ``halo_xray`` grids (with
    USE_MINI_HALOS also ``halo_sfr_mini`` and the scalar ``log10_Mcrit_MCG_ave``).
    The process-global parameters must have been broadcast to the library.  Returns a dict with
    ``filtered_sfr``, ``filtered_xray`` [N_STEP_TS, ...] and ``mean_sfr`` [N_STEP_TS] (mini-halos:
    ``filtered_sfr_mini``, ``mean_log10_Mcrit_LW`` and, under LYA_MULTIPLE_SCATTERING, the
    straight-line copies ``filtered_sfr_lw`` / ``filtered_sfr_mini_lw``)."""
    lib = lib or load(require_gpu=True)
    so, cp, ap, ao = simulation_options, cosmo_params, astro_params, astro_options
    mini = bool(ao.USE_MINI_HALOS)
    n_step = ap.N_STEP_TS
    shape = tuple(hboxes[0]["halo_sfr"].shape)
    on_device = hasattr(hboxes[0]["halo_sfr"], "is_cuda")  # torch history: zero-copy entry points
    if on_device:
        import torch

        dev = hboxes[0]["halo_sfr"].device

        def grid_stack():
            return torch.zeros((n_step,) + shape, dtype=torch.float32, device=dev)

        def fptr(a):
            return C.cast(a.data_ptr(), S.c_float_p)

        def all_zero(a):
            return not bool(torch.any(a != 0))
    else:
        def grid_stack():
            return np.zeros((n_step,) + shape, np.float32)

        def fptr(a):
            return a.ctypes.data_as(S.c_float_p)

        def all_zero(a):
            return bool(np.all(a == 0))
    cosmo = FlatCosmology(cp.hlittle, cp.OMm)
    R_range, zpp_avg = xray_shells(redshift, so.HII_DIM, so.BOX_LEN, n_step, ap.R_MAX_TS, cosmo)
    z_max = min(max(z_halos), so.Z_HEAT_MAX)
    if ao.LYA_MULTIPLE_SCATTERING:
        x_HI = 1.0 if previous_xHI_mean is None else float(previous_xHI_mean)
        R_star = lya_diffusion_scale(redshift, x_HI, cp.hlittle, cp.OMm, cp.OMb, cp.Y_He, cosmo)
    else:
        R_star = 0.0
    box = {"filtered_sfr": grid_stack(), "filtered_xray": grid_stack(),
           "mean_sfr": np.zeros(n_step), "mean_sfr_mini": np.zeros(n_step),
           "mean_log10_Mcrit_LW": np.zeros(n_step)}
    extra = {}
    if mini:
        names = ["filtered_sfr_mini"] + (["filtered_sfr_lw", "filtered_sfr_mini_lw"]
                                         if ao.LYA_MULTIPLE_SCATTERING else [])
        for k in names:
            box[k] = grid_stack()
            extra[k] = fptr(box[k])
    src = S.XraySourceBoxStruct(
        filtered_sfr=fptr(box["filtered_sfr"]), filtered_xray=fptr(box["filtered_xray"]), **extra,
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
            if mini:  # "minimum" (single_field.py:591; upstream's M_TURN is the log10 there)
                box["mean_log10_Mcrit_LW"][i] = math.log10(ap.M_TURN)
            continue
        fields = ("halo_sfr", "halo_xray") + (("halo_sfr_mini", "log10_Mcrit_MCG_ave") if mini else ())
        hb = interp_halo_boxes(z_sorted, b_sorted, fields, float(zpp_avg[i]))
        if all_zero(hb["halo_sfr"]) and (not mini or all_zero(hb["halo_sfr_mini"])):
            if mini:
                box["mean_log10_Mcrit_LW"][i] = hb["log10_Mcrit_MCG_ave"]
            continue
        hbs = S.HaloBoxStruct(halo_sfr=fptr(hb["halo_sfr"]), halo_xray=fptr(hb["halo_xray"]))
        if mini:
            hbs.halo_sfr_mini = fptr(hb["halo_sfr_mini"])
            hbs.log10_Mcrit_MCG_ave = float(hb["log10_Mcrit_MCG_ave"])
        check(lib.UpdateXraySourceBox(C.byref(hbs), R_inner, R_outer, i, R_star, C.byref(src)),
              "UpdateXraySourceBox")
    box["zpp_avg"], box["R_range"], box["R_star"] = zpp_avg, R_range, R_star
    return box


# =============================================================================================
# The evolution loop (reference: src/py21cmfast/drivers/coeval.py:749-890 `_redshift_loop_generator`
# with `run_coeval`'s set-up :560-745, and drivers/_global_initialization.py for the C state).
# =============================================================================================
TS_FIELDS = ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction")
ION_FIELDS = ("neutral_fraction", "z_reion", "kinetic_temperature", "unnormalised_nion",
              "ionisation_rate_G12", "mean_free_path", "cumulative_recombinations")


def ionisation_radii(so, ap, lagrangian: bool, ionise_entire_sphere: bool = False) -> int:
    """Number of filter radii of the excursion set (setup_radii, IonisationBox.c:964-1006): the
    length of IonizedBox.unnormalised_nion[_mini] with USE_MINI_HALOS."""
    L_FACTOR = 0.620350491
    pixel = float(so.BOX_LEN) / float(so.HII_DIM)
    r_max = min(float(ap.R_BUBBLE_MAX), L_FACTOR * float(so.BOX_LEN))
    # setup_radii, IonisationBox.c:968-972: the unit cell factor only without IONISE_ENTIRE_SPHERE
    cell_factor = 1.0 if (lagrangian and pixel < 1 and not ionise_entire_sphere) else L_FACTOR
    r_min = max(float(ap.R_BUBBLE_MIN), cell_factor * pixel)
    n_radii = int(math.log(r_max / r_min) / math.log(float(ap.DELTA_R_HII_FACTOR)) + 1)
    for i in range(n_radii):
        if r_min * float(ap.DELTA_R_HII_FACTOR) ** i > r_max - 1e-7:
            return i + 1
    return n_radii


def get_logspaced_redshifts(min_redshift, z_step_factor, max_redshift):
    """The node redshifts of an evolution, descending (wrapper/inputs.py:1774-1789)."""
    z = 10 ** np.arange(np.log10(1 + min_redshift), np.log10((1 + max_redshift) * z_step_factor),
                        np.log10(z_step_factor)) - 1
    return tuple(float(v) for v in z[::-1])


class Inputs:
    """The six parameter structs of a run (InputParameters, wrapper/inputs.py) with the defaults
    of ``structs.default_*``; keyword arguments are routed to the struct that has the field."""

    def __init__(self, random_seed=1, cosmo_tables=None, **kw):
        groups = (("simulation_options", S.SimulationOptions, S.default_simulation_options),
                  ("matter_options", S.MatterOptions, S.default_matter_options),
                  ("cosmo_params", S.CosmoParams, S.default_cosmo_params),
                  ("astro_params", S.AstroParams, S.default_astro_params),
                  ("astro_options", S.AstroOptions, S.default_astro_options))
        for name, cls, make in groups:
            names = {f[0] for f in cls._fields_}
            setattr(self, name, make(**{k: kw.pop(k) for k in list(kw) if k in names}))
        if kw:
            raise TypeError(f"unknown parameters: {sorted(kw)}")
        self.cosmo_tables = cosmo_tables or S.default_cosmo_tables()
        self.random_seed = int(random_seed)

    @property
    def evolution_required(self):
        """Whether a box depends on the previous snapshot (wrapper/inputs.py:1805-1815)."""
        return bool(self.astro_options.USE_TS_FLUCT or self.astro_options.RECOMB_MODEL != 0)

    def node_redshifts(self, out_redshifts):
        so = self.simulation_options
        if not self.evolution_required:
            return tuple(sorted((float(z) for z in out_redshifts), reverse=True))
        return get_logspaced_redshifts(min(out_redshifts), so.ZPRIME_STEP_FACTOR, so.Z_HEAT_MAX)



def required_redshifts(inputs: "Inputs", out_redshifts):
    """_get_required_redshifts_coeval (coeval.py:971-992): the node redshifts above the lowest
    requested one plus the requested redshifts themselves, descending and unique (float32 values,
    as the C entry points take them).  A requested redshift that is not a node is an extra
    snapshot computed at exactly that z; it does NOT become the next snapshot's "previous" box
    and its halo grids do not enter the history (coeval.py:880-884).  Returns (all_redshifts,
    set of those that are nodes)."""
    outs = [float(np.float32(z)) for z in out_redshifts]
    nodes = [float(np.float32(z)) for z in inputs.node_redshifts(outs)]
    if not inputs.evolution_required:
        allz = sorted(set(outs), reverse=True)
        return allz, set(allz)
    nodes = [z for z in nodes if z > min(outs)]
    return sorted(set(nodes) | set(outs), reverse=True), set(nodes)


def _initialise(lib, inputs: Inputs, data_path):
    """What GlobalInitializationManager does before the first Compute* call."""
    i = inputs
    lib.Broadcast_struct_global_all(*(C.byref(x) for x in (
        i.simulation_options, i.matter_options, i.cosmo_params, i.astro_params, i.astro_options,
        i.cosmo_tables)))
    if data_path is not None:
        i._data_path = str(data_path).encode()  # kept alive: C holds the pointer
        S.ConfigSettings.in_dll(lib, "config_settings").external_table_path = i._data_path
    lib.init_ps()
    if i.astro_options.USE_TS_FLUCT:
        lib.init_heat.restype = C.c_int
        if lib.init_heat() != 0:
            check(1, "init_heat")
    elif data_path is not None:
        lib.c21_recfast_load.restype = C.c_int
        check(lib.c21_recfast_load(), "recfast")
    if i.astro_options.RECOMB_MODEL != 0:
        lib.init_MHR.restype = None
        lib.init_MHR()


def run_coeval(inputs: Inputs, out_redshifts, *, data_path=None, device=None, lib=None,
               keep=("density", "velocity_z", "neutral_fraction", "z_reion", "brightness_temp") + TS_FIELDS,
               progress=None, halo_catalogs=None, inspect=None):
    """Evolve boxes through the library's entry points, mirroring ``run_coeval``: initial
    conditions once, then from the highest node redshift down: PerturbedField -> [HaloBox ->
    XraySourceBox ->] [TsBox ->] IonizedBox -> BrightnessTemp, every snapshot receiving the
    previous one's boxes.  Supported source models: CONST-ION-EFF, E-INTEGRAL, L-INTEGRAL and,
    with catalogues from the caller, DEXM-ESF / CHMF-SAMPLER: ``halo_catalogs(z)`` returns the
    catalogue of node redshift z as ``structs.HaloCatalogStruct`` (``structs.halo_catalog``; numpy or
    device arrays) -- finding and sampling halos is not part of this backend.

    ``device``: a torch device string ("cuda") keeps every array in HBM (zero-copy entry points);
    None uses numpy arrays that the library stages.  ``data_path``: directory of the reference's
    data tables (py21cmfast's ``_data``).  Returns ``{redshift: {field: array}}`` for the requested
    redshifts (fields in ``keep``; plus the scalars ``mean_f_coll`` and ``Q_HI``), keyed by the
    requested redshift as float32 -- a requested redshift between two nodes is computed AT that
    redshift from the last node above it, like upstream -- and, under the key ``"history"``, the
    global signal (z, mean dT_b, mean x_HI, mean T_s) of every snapshot computed.
    ``inspect(z, ctx)``: test hook called after every snapshot with the structs its
    ComputeIonizedBox call was given (``ctx["new_ion"]()`` allocates another output box)."""
    lib = lib or load(require_gpu=True)
    from . import grid_api as api

    so, mo, ao, ap = (inputs.simulation_options, inputs.matter_options, inputs.astro_options,
                      inputs.astro_params)
    if mo.SOURCE_MODEL in (3, 4) and halo_catalogs is None:
        raise NotImplementedError("SOURCE_MODEL = DEXM-ESF / CHMF-SAMPLER needs halo_catalogs(z): the halo "
                                  "finder and sampler are not part of this backend")
    mini = bool(ao.USE_MINI_HALOS)
    if mini and not (mo.SOURCE_MODEL in (1, 2, 3, 4) and ao.USE_TS_FLUCT):
        raise NotImplementedError("USE_MINI_HALOS runs with SOURCE_MODEL = E-INTEGRAL or L-INTEGRAL "
                                  "and USE_TS_FLUCT (the Lyman-Werner background comes from the TsBox)")
    _initialise(lib, inputs, data_path)
    n, nz = so.HII_DIM, int(so.NON_CUBIC_FACTOR * so.HII_DIM)
    shape = (n, n, nz)
    lagrangian, ts_on, recomb = mo.SOURCE_MODEL >= 2, bool(ao.USE_TS_FLUCT), ao.RECOMB_MODEL
    n_radii = ionisation_radii(so, ap, lagrangian, bool(ao.IONISE_ENTIRE_SPHERE))
    if device is not None:
        import torch

        def new(fill=0.0, shp=shape):
            return torch.full(shp, fill, dtype=torch.float32, device=device)

        def host(a):
            return a.cpu().numpy()
    else:
        def new(fill=0.0, shp=shape):
            return np.full(shp, fill, np.float32)

        def host(a):
            return a
    fp = api._fptr

    spec = S.IcsSpec(dim=so.DIM, dim_z=int(so.NON_CUBIC_FACTOR * so.DIM), hii_dim=n, hii_dim_z=nz,
                     perturb_algorithm=mo.PERTURB_ALGORITHM, perturb_on_high_res=int(mo.PERTURB_ON_HIGH_RES))
    ics = api.new_ics_arrays(spec, device=device)
    if mo.V_CB_MODEL == 1:
        ics["lowres_vcb"] = new()
    icss = api.ics_struct(ics)
    check(lib.ComputeInitialConditions(inputs.random_seed, C.byref(icss)), "ComputeInitialConditions")

    lib.ComputeTsBox.restype = C.c_int
    lib.ComputeTsBox.argtypes = [C.c_float, C.c_float, C.c_float, C.c_short] + [C.c_void_p] * 5
    lib.ComputeHaloBox.restype = C.c_int
    lib.ComputeHaloBox.argtypes = [C.c_double] + [C.c_void_p] * 5
    lib.ComputeBrightnessTemp.restype = C.c_int
    lib.ComputeBrightnessTemp.argtypes = [C.c_float] + [C.c_void_p] * 4

    def new_ion():
        rshape = shape if recomb != 1 else (1, 1, 1)
        arr = {k: new(1.0 if k == "neutral_fraction" else 0.0,
                      rshape if k == "cumulative_recombinations" else shape) for k in ION_FIELDS}
        if mini and not lagrangian:  # one f_coll grid per radius and population (outputs.py:1538-1543)
            arr["unnormalised_nion"] = new(0.0, (n_radii,) + shape)
            arr["unnormalised_nion_mini"] = new(0.0, (n_radii,) + shape)
        if mo.MINIMIZE_MEMORY:
            arr.pop("kinetic_temperature"), arr.pop("mean_free_path")
        return arr, S.IonizedBoxStruct(**{k: fp(v) for k, v in arr.items()})

    def new_ts():
        arr = {k: new() for k in TS_FIELDS + (("J_21_LW",) if mini else ())}
        return arr, S.TsBoxStruct(**{k: fp(v) for k, v in arr.items()})

    out_redshifts = [float(np.float32(z)) for z in out_redshifts]
    all_redshifts, is_node = required_redshifts(inputs, out_redshifts)
    wanted = set(out_redshifts)
    prev_ion_arr, prev_ion = new_ion()
    prev_ts_arr, prev_ts = new_ts()
    prev_pf_arr = None
    prev_z, prev_xHI = 0.0, None
    prev_means = (0.0, 0.0)
    z_halos, hboxes = [], []
    result, history = {}, []
    for z in all_redshifts:
        pf_arr = {"density": new(), "velocity_z": new()}
        pf = S.PerturbedFieldStruct(**{k: fp(v) for k, v in pf_arr.items()})
        check(lib.ComputePerturbedField(z, C.byref(icss), C.byref(pf)), "ComputePerturbedField")
        hb_arr, hb = {}, S.HaloBoxStruct()
        if lagrangian:
            names = (["n_ion", "halo_sfr"] + (["halo_xray"] if ts_on else [])
                     + (["whalo_sfr"] if recomb else []) + (["halo_sfr_mini"] if mini else []))
            hb_arr = {k: new() for k in names}
            hb = S.HaloBoxStruct(**{k: fp(v) for k, v in hb_arr.items()})
            # (mini-halos: turnover masses from the previous snapshot's J_21_LW, Gamma_12, z_reion)
            cat = halo_catalogs(z) if mo.SOURCE_MODEL in (3, 4) else None
            check(lib.ComputeHaloBox(z, C.byref(icss), C.byref(cat) if cat is not None else None,
                                     C.byref(prev_ts) if mini else None,
                                     C.byref(prev_ion) if mini else None, C.byref(hb)),
                  "ComputeHaloBox")
        ts_arr, ts = ({}, S.TsBoxStruct())
        if ts_on:
            srcs = None
            if lagrangian:  # the X-ray light cone reads the halo-grid HISTORY (host arrays)
                # (device runs keep the history in HBM: interpolation and filtering never leave it)
                hist = {k: hb_arr[k] for k in ("halo_sfr", "halo_xray")
                        + (("halo_sfr_mini",) if mini else ())}
                if mini:
                    hist["log10_Mcrit_MCG_ave"] = hb.log10_Mcrit_MCG_ave
                xsrc = compute_xray_source_field(
                    z_halos + [z], hboxes + [hist], z, simulation_options=so,
                    cosmo_params=inputs.cosmo_params, astro_params=ap, astro_options=ao,
                    previous_xHI_mean=prev_xHI, lib=lib)
                srcs = S.XraySourceBoxStruct(filtered_sfr=fp(xsrc["filtered_sfr"]),
                                             filtered_xray=fp(xsrc["filtered_xray"]))
                if mini:
                    for k in ("filtered_sfr_mini", "filtered_sfr_lw", "filtered_sfr_mini_lw"):
                        if k in xsrc:
                            setattr(srcs, k, fp(xsrc[k]))
                    srcs.mean_log10_Mcrit_LW = xsrc["mean_log10_Mcrit_LW"].ctypes.data_as(
                        C.POINTER(C.c_double))
                if z in is_node:  # hbox_arr grows on the nodes only (coeval.py:880-884)
                    z_halos.append(z)
                    hboxes.append(hist)
            ts_arr, ts = new_ts()
            check(lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), C.byref(srcs) if srcs else None,
                                   C.byref(prev_ts), C.byref(icss), C.byref(ts)), "ComputeTsBox")
        ion_arr, ion = new_ion()
        if prev_pf_arr is None and mini:
            # the first snapshot's "previous" field is a dummy that ComputeIonizedBox overwrites
            # with -1.5 (IonisationBox.c:394-398): it must not alias the current density
            prev_pf_arr = {"density": new(), "velocity_z": new()}
        if mini:  # the trapezoidal means live in the structs (set_mean_fcoll, :476-501)
            prev_ion.mean_f_coll, prev_ion.mean_f_coll_MINI = prev_means
        prev_pf = S.PerturbedFieldStruct(**{k: fp(v) for k, v in (prev_pf_arr or pf_arr).items()})
        check(lib.ComputeIonizedBox(z, prev_z, C.byref(pf), C.byref(prev_pf), C.byref(prev_ion),
                                    C.byref(ts), C.byref(hb), C.byref(icss), C.byref(ion)),
              "ComputeIonizedBox")
        bt_arr = {"brightness_temp": new(), "tau_21": new()}
        bt = S.BrightnessTempStruct(**{k: fp(v) for k, v in bt_arr.items()})
        check(lib.ComputeBrightnessTemp(z, C.byref(ts), C.byref(ion), C.byref(pf), C.byref(bt)),
              "ComputeBrightnessTemp")
        mean = lambda a: float(a.double().mean()) if device is not None else float(a.mean(dtype=np.float64))  # noqa: E731
        prev_xHI = mean(ion_arr["neutral_fraction"])
        history.append((z, mean(bt_arr["brightness_temp"]), prev_xHI,
                        mean(ts_arr["spin_temperature"]) if ts_on else float("nan")))
        if progress:
            progress(history[-1])
        if inspect:  # test hook: the structs of this snapshot's ComputeIonizedBox call, before they age
            inspect(z, dict(prev_z=prev_z, pf=pf, prev_pf=prev_pf, prev_ion=prev_ion, ts=ts, hb=hb,
                            icss=icss, ion=ion, ion_arr=ion_arr, ts_arr=ts_arr, pf_arr=pf_arr,
                            new_ion=new_ion))
        if z in wanted:
            boxes = {**pf_arr, **hb_arr, **ts_arr, **ion_arr, **bt_arr}
            snap = {k: boxes[k] for k in keep if k in boxes}
            snap["mean_f_coll"], snap["Q_HI"] = ion.mean_f_coll, ts.Q_HI
            if mini:
                snap["mean_f_coll_MINI"] = ion.mean_f_coll_MINI
                snap["log10_Mturnover_ave"] = ion.log10_Mturnover_ave
                snap["log10_Mturnover_MINI_ave"] = ion.log10_Mturnover_MINI_ave
            result[z] = snap
        if z in is_node:
            prev_means = (ion.mean_f_coll, ion.mean_f_coll_MINI)
        if inputs.evolution_required and z in is_node:  # only nodes are the next one's "previous"
            prev_ts_arr, prev_ts, prev_ion_arr, prev_ion, prev_pf_arr, prev_z = (
                ts_arr, ts, ion_arr, ion, pf_arr, z)
    result["history"] = history
    result["initial_conditions"] = ics
    return result
