"""GPU: ComputeHaloBox with USE_MINI_HALOS against the oracle -- the turnover grids (with
upstream's per-thread running maximum of the atomic turnover), the four-value deposit with 2-D
table lookups, and the entry point.  Tolerances as tests/test_gpu_halobox.py."""
import ctypes as C
import importlib
import math

import numpy as np
import pytest

import halobox_mini_helpers as HM
from test_gpu_halobox import api, compare  # noqa: F401  (fixture)
from test_oracle_halobox import random_ics

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.mark.parametrize("n,lpt2,vscale,device,xray", [(16, 1, 1.0, False, True), (32, 1, 6.0, True, True),
                                                       (24, 0, 20.0, True, False), (33, 1, 3.0, False, True)])
def test_deposit_matches_oracle(api, oracle, n, lpt2, vscale, device, xray):
    spec = HM.mini_spec(n, lpt2=lpt2, xray=xray)
    ics = random_ics(n, n, False, seed=n, vscale=vscale)
    ref = oracle.halobox_grids(spec, ics, with_whalo=True, with_xray=xray)
    if device:
        import torch

        ics = {k: torch.from_numpy(v).cuda() for k, v in ics.items()}
    got = api.halobox_grids(spec, ics, with_whalo=True, with_xray=xray)
    assert set(got) == set(ref) and "halo_sfr_mini" in got
    compare(got, ref)
    assert ref["halo_sfr_mini"].max() > 0


@pytest.mark.parametrize("threads", [1, 5, 16])
def test_turnovers_match_oracle(api, oracle, threads):
    shape = (20, 20, 28)
    spec, g12, zre, j21, vcb = HM.turnover_inputs(shape)
    for v in (vcb, None):
        ra, rm, rave = oracle.halobox_turnovers(spec, 1e5, 1, threads, g12, zre, j21, v)
        a, m, ave = api.halobox_turnovers(spec, 1e5, 1, threads, g12, zre, j21, v)
        np.testing.assert_allclose(a, ra, rtol=3e-7)
        np.testing.assert_allclose(m, rm, rtol=3e-7)
        assert ave == pytest.approx(rave, rel=1e-7)
    import torch
    dev = lambda x: torch.from_numpy(x).cuda()  # noqa: E731
    a, m, ave = api.halobox_turnovers(spec, 1e5, 1, threads, dev(g12), dev(zre), dev(j21), dev(vcb))
    ra, rm, rave = oracle.halobox_turnovers(spec, 1e5, 1, threads, g12, zre, j21, vcb)
    np.testing.assert_allclose(a.cpu().numpy(), ra, rtol=3e-7)
    # above Z_HEAT_MAX: no inputs needed
    a0, m0, _ = api.halobox_turnovers(spec, 1e5, 0, threads, None, None, None, vcb, like=vcb)
    r0 = oracle.halobox_turnovers(spec, 1e5, 0, threads, None, None, None, vcb, shape=shape)
    np.testing.assert_allclose(a0, r0[0], rtol=3e-7)
    np.testing.assert_allclose(m0, r0[1], rtol=3e-7)


def test_high_resolution_sources_are_refused(api):
    from test_oracle_halobox import halobox_spec, make_tables
    spec = HM.add_minis(halobox_spec(16, 32, True, make_tables()), 16)
    ics = random_ics(16, 32, True, seed=1)
    with pytest.raises(RuntimeError, match="PERTURB_ON_HIGH_RES"):
        api.halobox_grids(spec, ics)
