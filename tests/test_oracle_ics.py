"""Pins the oracle's ComputeInitialConditions restatement.

* The reference's round-trip test (tests/test_initial_conditions.py:153-167): ICs regenerated
  from their own hires_density (the `initial_density` path, InitialConditions.c:620-663)
  reproduce every low-res field to atol 1e-5.
* Independent numpy check of the k-space operators: with V = L^3 and the reference's
  conventions, lowres_v = Re IFFT(i k_a / k^2 * delta_k) evaluated with numpy.fft.
* Hermitian symmetry: the sampled hires density field is real-consistent
  (r2c of the c2r output reproduces delta_k on the k_z = 0 / Nyquist planes).
* Gaussianity of the Philox + Box-Muller stream.
"""

import ctypes as C
import importlib

import numpy as np
import pytest

S = importlib.import_module("21cmfast_amd.structs")


def ics_spec(dim, hii_dim, box_len, algorithm=2, hires=0, seed=1234, density_is_input=0,
             index=-2.0, amp=30.0):
    n_m = 3 * (dim // 2) ** 2 + 1
    dk = 2 * np.pi / box_len
    m = np.arange(n_m, dtype=np.float64)
    k = dk * np.sqrt(m)
    pk = np.zeros(n_m)
    pk[1:] = amp * k[1:] ** index
    pk = np.ascontiguousarray(pk)
    vol = np.float32(np.float32(box_len) * np.float32(box_len)) * np.float32(1.0)
    vol = np.float32(vol * np.float32(box_len))
    spec = S.IcsSpec(dim=dim, dim_z=dim, hii_dim=hii_dim, hii_dim_z=hii_dim, box_len=box_len,
                     box_len_z=box_len, volume=float(vol), perturb_algorithm=algorithm,
                     perturb_on_high_res=hires, density_is_input=density_is_input, n_m=n_m,
                     pk_by_m=pk.ctypes.data_as(S.c_double_p), seed=seed)
    spec._pk = pk  # keep alive
    return spec


LOWRES_FIELDS = ["lowres_density", "lowres_vx", "lowres_vy", "lowres_vz", "lowres_vx_2LPT",
                 "lowres_vy_2LPT", "lowres_vz_2LPT"]


@pytest.mark.parametrize("dim,hii_dim", [(24, 8), (32, 16), (20, 10)])
def test_roundtrip_from_own_density(oracle, dim, hii_dim):
    spec = ics_spec(dim, hii_dim, box_len=3.0 * hii_dim)
    ic = oracle.ics_grids(spec)
    assert ic["hires_density"].std() > 0
    spec2 = ics_spec(dim, hii_dim, box_len=3.0 * hii_dim, density_is_input=1)
    start = oracle.new_ics_arrays(spec2)
    start["hires_density"][...] = ic["hires_density"]
    ic2 = oracle.ics_grids(spec2, start)
    assert np.all(ic2["hires_density"] == ic["hires_density"])
    for name in LOWRES_FIELDS:
        scale = max(1.0, np.abs(ic[name]).max())
        np.testing.assert_allclose(ic[name], ic2[name], atol=1e-5 * scale, rtol=0.0,
                                   err_msg=name)


def test_velocities_against_numpy(oracle):
    """DIM == HII_DIM (no filtering, no subsampling): v_a = IFFT(i k_a/k^2 delta_k)."""
    n, L = 16, 40.0
    spec = ics_spec(n, n, L, algorithm=2)
    ic = oracle.ics_grids(spec)
    delta = ic["hires_density"].astype(np.float64)
    np.testing.assert_allclose(ic["lowres_density"], ic["hires_density"], atol=1e-6)
    # half-spectrum + irfftn == the reference's r2c / c2r semantics (x,y complex, z real)
    dk = np.fft.rfftn(delta)
    kf = 2 * np.pi * np.fft.fftfreq(n, d=L / n)
    kf[n // 2] = np.pi * n / L  # index_to_k keeps the Nyquist index positive (idx <= dim/2)
    kzf = 2 * np.pi * np.fft.rfftfreq(n, d=L / n)
    kx, ky, kz = np.meshgrid(kf, kf, kzf, indexing="ij")
    k2 = kx**2 + ky**2 + kz**2
    k2[0, 0, 0] = 1.0
    shape = (n, n, n)
    for ax, kk in zip("xyz", (kx, ky, kz)):
        v = np.fft.irfftn(1j * kk / k2 * dk, s=shape, axes=(0, 1, 2))
        np.testing.assert_allclose(ic["lowres_v" + ax], v, atol=2e-5 * np.abs(v).max())
    # 2LPT source: sum_{i<j} (phi_ii phi_jj - phi_ij^2), phi_ij = IFFT(-k_i k_j/k^2 delta_k)
    ks = (kx, ky, kz)
    phi = {(a, b): np.fft.irfftn(-ks[a] * ks[b] / k2 * dk, s=shape, axes=(0, 1, 2))
           for a in range(3) for b in range(a, 3)}
    src = sum(phi[(a, a)] * phi[(b, b)] - phi[(a, b)] ** 2 for a, b in ((0, 1), (0, 2), (1, 2)))
    sk = np.fft.rfftn(src)
    for ax, kk in zip("xyz", (kx, ky, kz)):
        v2 = np.fft.irfftn(1j * kk / k2 * sk, s=shape, axes=(0, 1, 2))
        np.testing.assert_allclose(ic[f"lowres_v{ax}_2LPT"], v2, atol=5e-5 * np.abs(v2).max())


def test_sampled_field_is_real_consistent(oracle):
    n, L = 16, 32.0
    spec = ics_spec(n, n, L, algorithm=1)
    ic = oracle.ics_grids(spec)
    # variance of delta matches sum P(k)/V over modes to sampling noise
    dk = 2 * np.pi / L
    f = np.fft.fftfreq(n, d=1.0 / n)
    mx, my, mz = np.meshgrid(f, f, f, indexing="ij")
    m = (mx**2 + my**2 + mz**2).astype(int)
    expected_var = spec._pk[m].sum() / L**3
    assert ic["hires_density"].var() == pytest.approx(expected_var, rel=0.15)
    assert abs(ic["hires_density"].mean()) < 1e-6


def test_gaussian_stream(oracle):
    a, b = zip(*(oracle.gaussian_pair(i, 42) for i in range(20000)))
    x = np.array(a + b)
    assert abs(x.mean()) < 0.02 and abs(x.std() - 1) < 0.02
    assert abs((x**4).mean() - 3) < 0.15
    assert oracle.gaussian_pair(7, 42) == oracle.gaussian_pair(7, 42)
    assert oracle.gaussian_pair(7, 42) != oracle.gaussian_pair(7, 43)


def vcb_table(dim, box_len, power=-0.6, amp=3.0e-3):
    """h(|k|) = sqrt(P_vcb / P) c_kms / |k| per |k|^2 index for a toy ratio P_vcb / P ~ k^power."""
    n_m = 3 * (dim // 2) ** 2 + 1
    k = 2 * np.pi / box_len * np.sqrt(np.arange(n_m, dtype=np.float64))
    h = np.zeros(n_m)
    h[1:] = amp * k[1:] ** (power / 2) * 2.99792458e5 / k[1:]
    return np.ascontiguousarray(h)


def test_relative_velocities_against_numpy(oracle):
    """compute_relative_velocities (reference: InitialConditions.c:141-238) without filtering or
    subsampling (DIM == HII_DIM): v_cb = sqrt(sum_a IFFT(i k_a h(k) delta_k)^2) / V x V."""
    n, L = 16, 40.0
    spec = ics_spec(n, n, L, algorithm=1)
    h = vcb_table(n, L)
    spec.vcb_by_m = h.ctypes.data_as(S.c_double_p)
    ics = oracle.new_ics_arrays(spec)
    ics["lowres_vcb"] = np.zeros((n, n, n), np.float32)
    ic = oracle.ics_grids(spec, ics)
    delta = ic["hires_density"].astype(np.float64)
    dk = np.fft.rfftn(delta)
    kf = 2 * np.pi * np.fft.fftfreq(n, d=L / n)
    kf[n // 2] = np.pi * n / L
    kzf = 2 * np.pi * np.fft.rfftfreq(n, d=L / n)
    kx, ky, kz = np.meshgrid(kf, kf, kzf, indexing="ij")
    f = np.fft.fftfreq(n, d=1.0 / n)
    mx, my, mz = np.meshgrid(np.abs(f), np.abs(f), np.arange(n // 2 + 1.0), indexing="ij")
    hk = h[(mx**2 + my**2 + mz**2).astype(int)]
    tot = 0.0
    for kk in (kx, ky, kz):
        v = np.fft.irfftn(1j * kk * hk * dk, s=(n, n, n), axes=(0, 1, 2))
        tot = tot + v * v
    want = np.sqrt(tot)
    np.testing.assert_allclose(ic["lowres_vcb"], want, rtol=2e-4, atol=2e-5 * want.max())
    assert want.mean() > 0
